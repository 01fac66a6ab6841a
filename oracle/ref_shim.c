/*
 * oracle/ref_shim.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Translation unit that turns the UNMODIFIED reference header into a shared
 * object.  No reference source is copied here: the header is #included from
 * where it lies (-I/root/reference) exactly the way the reference's own wasm
 * example overrides the GS_API seam (examples/wasm/grayskull.c:31-34).
 *
 * Build recipe: oracle/Makefile -> oracle/_ref/libgs_ref.so (git-ignored).
 * Flags mirror the reference Makefile:1 (-std=c99 => no FP contraction).
 */
#define GS_API /* external linkage for every GS_API function (grayskull.h:7-9) */
#include "grayskull.h"
#include "examples/nanomagick/frontalface.h"

/* Static helpers of the reference are reachable only through wrappers. */
unsigned ref_integral_sum(const unsigned *ii, unsigned iw, unsigned x, unsigned y, unsigned w,
                          unsigned h) {
  return gs_integral_sum(ii, iw, x, y, w, h); /* grayskull.h:754 */
}
int ref_lbp_code(const unsigned *ii, unsigned iw, int x, int y, int fx, int fy, int fw, int fh) {
  return gs_lbp_code(ii, iw, x, y, fx, fy, fw, fh); /* grayskull.h:769 */
}
unsigned ref_hamming_distance(const uint32_t a[8], const uint32_t b[8]) {
  return gs_hamming_distance(a, b); /* grayskull.h:671 */
}
void ref_sort_keypoints(struct gs_keypoint *kps, unsigned n) {
  if (n > 1) gs_sort_keypoints(kps, n); /* grayskull.h:639 (guard as in :657) */
}
const struct gs_lbp_cascade *ref_frontalface(void) { return &frontalface; }
const int *ref_brief_pattern(void) { return &gs_brief_pattern[0][0]; } /* grayskull.h:541 */
float ref_atan2(float y, float x) { return gs_atan2(y, x); }           /* grayskull.h:100 */
float ref_sin(float x) { return gs_sin(x); }                           /* grayskull.h:101 */
