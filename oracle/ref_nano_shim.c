/*
 * oracle/ref_nano_shim.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * The reference's pyramid ORB driver lives in its CLI (examples/nanomagick/nanomagick.c:245-290,
 * `static`), not in the header.  This translation unit #includes the UNMODIFIED CLI source from
 * where it lies (its main() renamed) and exports one wrapper around that function, so the oracle's
 * restatement (orc_orb_extract_pyramid) can be pinned against it.  Nothing is copied.
 * Build recipe: oracle/Makefile -> oracle/_ref/libgs_ref_nano.so (git-ignored).
 */
#define main nanomagick_cli_main
#include "examples/nanomagick/nanomagick.c"
#undef main

unsigned ref_extract_pyramid_orb(struct gs_image img, struct gs_keypoint *kps, unsigned nkps,
                                 unsigned threshold, uint8_t *buffer, unsigned n_levels) {
  return extract_pyramid_orb_nm(img, kps, nkps, threshold, buffer, n_levels); /* nanomagick.c:245 */
}
