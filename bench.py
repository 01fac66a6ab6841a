#!/usr/bin/env python3
"""bench.py -- throughput of the Grayskull hot path on MI355X (driver contract: one JSON line).

Workload (BASELINE.json configs[1]): per frame gs_blur(r=2) -> gs_sobel (into a zeroed image) ->
gs_otsu_threshold -> gs_threshold on 3840x2160 uint8, over a device-resident batch of
synthetic block-noise frames (SURVEY.md 8c generator, run on the GPU, bit-identical to the CPU
one).  A step = one pass of that chain over the whole batch; value = input Mpix/s, whole job.
Frames shard by frame across ranks (weak scaling: --frames per GPU); no data-path collective.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment launches the N ranks
itself (torch.distributed.run, one process per GPU, RCCL); under the driver's own torchrun launch it
is simply one of the ranks.

roofline: the dominant kernel of the timed step (the fused blur+sobel+histogram kernel), timed live
with HIP events on the stream it is launched on.  `frac` is PHYSICAL: the bytes the launch moves
(1 read + 1 write per pixel; PMC traffic beside it) / time / 8 TB/s.  The kernel is VALU-bound, so
the `valu` block prices its instruction mix against the issue rates measured by
scripts/ubench_valu.cpp (profiles/r02a_ubench_valu.log).  SURVEY 8(d)'s per-call-equivalent figure
(5 B/px for the three calls it replaces) is reported as `percall_equivalent`, never as `frac`.
cpu_baseline: the unmodified reference (oracle/_ref, kind "reference") or the C restatement
(kind "port") on the host cores, same chain, bounded frame sample, rank 0 at N=1 only.
"""
import argparse
import json
import os
import socket
import sys
import time

# the host driver only supports dmabuf IPC: without this RCCL fails with `hipIpcGetMemHandle: invalid argument` (exported on the
# GPU boxes already; set here too so that every way of launching the ranks has it before torch / HIP start)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
NORTH_STAR_TARGET_FRAC = 0.6  # BASELINE.json north_star: ">= 60 % of MI355X HBM-read roofline on gs_sobel at 4Kx4K" (a constant, never measured)
HBM_COPY_GBS = 6290.0   # same table: measured float4 copy ceiling
BASELINE_METRIC = "Mpix/s (and % HBM roofline) for gs_sobel+gs_blur on 4K uint8, 1/2/4/8 GPU"

# Fused kernel k_blur_sobel_hist16<2>: VALU wave-instructions per wave per source row (16 px per lane),
# by issue class, counted in the source and confirmed against the ISA by
# scripts/isa_count.py; rates = chip-wide wave-instructions/s measured by scripts/ubench_valu.cpp at
# 8 waves per SIMD (profiles/r02a_ubench_valu.log).  "full" = v_add_u32 / v_sub_u32 / v_and_b32 ...,
# "half" = packed-16, v_perm_b32, v_alignbit_b32, multiplies, every VOP3 form.
VALU_RATE_GINST = {"full": 1000.0, "half": 578.0}
FUSED_OPS_PER_ROW_FILE = os.path.join(ROOT, "profiles", "fused_isa_mix.json")


def time_stream(torch, fn, reps, warm=1):
    """average ms per call of fn(), events recorded on the (shared) launch stream"""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


LDS_ATOMIC_TLANE_S = 9.4  # ds_add_u32, lane-private copies, any number of active lanes: 148 G wave-inst/s (r02i_ubench_new_ops.log)


def fast_valu_block(npx_launch, ms_score):
    """VALU issue pricing of the gs_fast score kernel from its SQ_INSTS_VALU count (profiles/fast_valu_pmc.json)"""
    d, stamp = measurement(os.path.join(ROOT, "profiles", "fast_valu_pmc.json"))
    if d is None:
        return {"stale_or_missing": stamp}
    insts = d["valu_wave_insts_per_px"] * npx_launch
    sec = insts * d["avg_issue_cycles"] / (1024 * 2.4e9)  # 1024 SIMDs, one VALU issue port each
    return {"valu_wave_insts_per_px": d["valu_wave_insts_per_px"], "avg_issue_cycles": d["avg_issue_cycles"],
            "score_kernel_ms": round(ms_score, 4), "valu_ms_at_issue_rate": round(sec * 1e3, 4),
            "valu_frac": round(sec * 1e3 / ms_score, 4), "source": d.get("source"), "taken_at": stamp}


def hbm_block(nbytes, ms, **extra):
    """physical HBM roofline block of one launch (or launch group): bytes really moved / time / peak"""
    gbs = nbytes / ms / 1e6
    d = {"ms": round(ms, 4), "bytes": float(nbytes), "GB/s": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4),
         "frac_of_copy_ceiling": round(gbs / HBM_COPY_GBS, 4)}
    d.update(extra)
    return d


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def physical_cores():
    """distinct (package, core) pairs among the CPUs this process may run on"""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1
    seen = set()
    for c in cpus:
        try:
            base = "/sys/devices/system/cpu/cpu%d/topology/" % c
            seen.add((open(base + "physical_package_id").read().strip(), open(base + "core_id").read().strip()))
        except OSError:
            return max(1, len(cpus) // 2)
    return max(1, len(seen))


def cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited: the GPU boxes
    show 256 hardware threads to a process that is allowed ~8 of them, and threads beyond the quota only take turns"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def cpu_baseline_pthreads(w, h, radius):
    """The unmodified reference behind a pthread loop over frames (oracle/ref_bench.c -> oracle/_ref/libgs_ref_bench.so):
    one run per thread count 1, 2, 4, ... while the aggregate keeps growing, buffers preallocated and touched, threads
    released together.  The best aggregate is `value`, its thread count `cores`."""
    import ctypes as C
    import numpy as np
    from oracle import pyoracle
    path = os.path.join(ROOT, "oracle", "_ref", "libgs_ref_bench.so")
    if not os.path.exists(path):
        return None
    lib = C.CDLL(path)
    lib.ref_chain_frames.restype = C.c_double
    lib.ref_chain_frames.argtypes = [C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_void_p, C.c_uint,
                                     C.POINTER(C.c_ulonglong)]
    nsrc = 8
    frames = np.stack([pyoracle.Oracle.synth(w, h, 1000 + i) for i in range(nsrc)])
    px = float(w * h)
    runs, t_all = [], time.time()
    chk = C.c_ulonglong(0)
    dt1 = lib.ref_chain_frames(1, 2, w, h, radius, frames.ctypes.data, nsrc, C.byref(chk))
    one = 2 * px / dt1 / 1e6
    runs.append({"threads": 1, "frames": 2, "seconds": round(dt1, 3), "Mpix/s": round(one, 2)})
    # thread counts double until a run takes 2.5 x as long as one frame on one thread: a container that shows 256 hardware
    # threads but is allowed ~8 CPUs' worth of time stops at 32 after sub-second runs instead of time-slicing 256 threads
    # for half a minute (one frame per thread: the threads start together, so one frame each is a fair sample)
    quota = cpu_quota()
    cap = usable_cores() if not quota else min(usable_cores(), 2 * max(1, int(round(quota))))
    nt, per_frame_s = 2, dt1 / 2
    while nt <= cap:
        dt = lib.ref_chain_frames(nt, 1, w, h, radius, frames.ctypes.data, nsrc, C.byref(chk))
        if dt <= 0:
            break
        v = nt * px / dt / 1e6
        runs.append({"threads": nt, "frames": nt, "seconds": round(dt, 3), "Mpix/s": round(v, 2),
                     "parallel_efficiency": round(v / (nt * one), 3)})
        if dt > 2.5 * per_frame_s:
            break
        nt = min(2 * nt, cap) if nt < cap else cap + 1
    best = max(runs, key=lambda r: r["Mpix/s"])
    return {"value": best["Mpix/s"], "unit": "Mpix/s", "cores": best["threads"], "kind": "reference",
            "seconds": round(time.time() - t_all, 2), "runs": runs,
            "single_thread": {"value": round(one, 2), "unit": "Mpix/s", "cores": 1, "frames": 2, "seconds": round(dt1, 3)},
            "cpu_model": cpu_model(), "nproc": os.cpu_count(), "usable_cores": usable_cores(), "physical_cores": physical_cores(),
            "cgroup_cpu_quota": quota,
            "sample": "%d frames %dx%d on %d threads (best of the thread counts in `runs`), blur(r=%d)->sobel->otsu->threshold, "
                      "unmodified reference C (oracle/_ref/libgs_ref_bench.so: grayskull.h as it lies, gcc -std=c99 -O2, pthread "
                      "loop over frames in the harness, buffers preallocated)" % (best["frames"], w, h, best["threads"], radius)}


def cpu_baseline(w, h, radius, frames_target=48):
    """SURVEY 8(d) / BASELINE.md 4: the unmodified reference on this box's host cores, (i) one thread -- the
    reference's native mode -- and (ii) all host cores, each thread looping over its own frames (the reference code
    itself stays single-threaded): the pthread harness when oracle/_ref has it, else Python threads over the ctypes
    oracle.  `value` / `cores` are the best all-cores figure."""
    try:
        d = cpu_baseline_pthreads(w, h, radius)
        if d:
            return d
    except Exception as e:  # a missing / foreign .so must not cost the bench line
        sys.stderr.write("cpu_baseline: pthread harness unavailable (%s), falling back to Python threads\n" % e)
    import threading
    from oracle import pyoracle
    kind = "reference" if pyoracle.have_reference() else "port"
    cores = usable_cores()
    per = max(1, frames_target // cores)  # the threads start together behind a barrier, so one frame each is a fair sample
    frames = per * cores
    imgs = [pyoracle.Oracle.synth(w, h, 1000 + i) for i in range(min(cores, 16))]  # shared, read-only
    oracles = [pyoracle.Oracle(kind) for _ in range(cores)]  # library handles made outside the timed region
    gate = threading.Barrier(cores + 1)

    def chain(o, img, n):  # ctypes releases the GIL inside the C calls
        for _ in range(n):
            s = o.sobel(o.blur(img, radius))
            o.threshold(s, o.otsu_threshold(s))

    def work(i):
        gate.wait()
        chain(oracles[i], imgs[i % len(imgs)], per)

    t0 = time.time()
    chain(oracles[0], imgs[0], 2)
    dt1 = time.time() - t0
    ths = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    for t in ths:
        t.start()
    gate.wait()  # every thread exists and waits: the clock starts with the work
    t0 = time.time()
    for t in ths:
        t.join()
    dt = time.time() - t0
    what = ("unmodified reference C (oracle/_ref, gcc -std=c99 -O2)" if kind == "reference"
            else "C restatement (oracle/gs_oracle.c)")
    return {"value": round(frames * w * h / dt / 1e6, 2), "unit": "Mpix/s", "cores": cores,
            "kind": kind, "seconds": round(dt + dt1, 2),
            "single_thread": {"value": round(2 * w * h / dt1 / 1e6, 2), "unit": "Mpix/s", "cores": 1, "frames": 2,
                              "seconds": round(dt1, 2)},
            "cpu_model": cpu_model(), "nproc": os.cpu_count(), "usable_cores": cores,
            "sample": "%d frames %dx%d, blur(r=%d)->sobel->otsu->threshold, %d Python threads x %d frames (+ 2 frames on one "
                      "thread), %s" % (frames, w, h, radius, cores, per, what)}


def measurement(path, only=None):
    """a committed measurement file of profiles/ plus whether the kernel sources it was taken from are still the
    ones in the tree (scripts/stamp.py): returns (dict or None, stamp) -- a STALE file is not used.
    only: the sources of the kernel whose row the caller uses (the file may hold rows of kernels from other sources)"""
    try:
        d = json.load(open(path))
    except Exception:
        return None, None
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from stamp import fresh
    ok = fresh(d, only)
    stamp = {"file": os.path.relpath(path, ROOT), "commit": d.get("stamped_at_commit"),
             "kernel_sources_unchanged": ok}
    if only is not None:
        stamp["sources_checked"] = list(only)
    return (None if ok is False else d), stamp


GOLDEN_BATCH = os.path.join(ROOT, "tests", "golden", "batch_checksums.json")


def load_golden_batch(w, h, r):
    """tests/golden/batch_checksums.json (reference-generated, tests/golden/make_batch_golden.py) when it covers this shape"""
    try:
        gold = json.load(open(GOLDEN_BATCH))
    except Exception:
        return None
    return gold if (gold["w"], gold["h"], gold["radius"]) == (w, h, r) else None


def frame_ranges(frames):
    """[0, 1, 2, 3, 7, 9, 10] -> '0-3, 7, 9-10' (the golden-checked frames of a share can be all 512 of them)"""
    out, fs = [], sorted(frames)
    i = 0
    while i < len(fs):
        j = i
        while j + 1 < len(fs) and fs[j + 1] == fs[j] + 1:
            j += 1
        out.append("%d" % fs[i] if i == j else "%d-%d" % (fs[i], fs[j]))
        i = j + 1
    return ", ".join(out)


def wsum_bytes(np, a):
    """sum (i+1)*(byte+1) mod 2^64 over raw bytes: the host twin of gsh_checksum_batch, for KB-sized result lists"""
    b = np.ascontiguousarray(a).view(np.uint8).reshape(-1).astype(np.uint64)
    return int(np.sum(np.arange(1, b.size + 1, dtype=np.uint64) * (b + np.uint64(1)), dtype=np.uint64))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks_if_needed(args):
    """`python bench.py --gpus N` on its own: become `torch.distributed.run` with N ranks."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    if os.environ.get("GS_BENCH_BACKEND") != "gloo":  # the gloo rehearsal shares one GPU on purpose
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.exit("bench.py: --gpus %d requested but only %d GPU(s) visible -- refusing to run a smaller job "
                     "under that label" % (args.gpus, have))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execv(sys.executable, cmd)


def rank_times(sh, torch, dt_local, steps):
    """every rank's own time for the timed steps (its clock stops at its own synchronize, before the closing barrier),
    gathered in rank order, with the skew the slowest rank imposes on the job"""
    t = sh.all_gather_frames(torch.tensor([dt_local / steps * 1e3], dtype=torch.float64), sh.world).tolist()
    return {"per_rank": [round(x, 4) for x in t], "max": round(max(t), 4), "min": round(min(t), 4),
            "skew_max_over_min": round(max(t) / min(t), 4) if min(t) > 0 else None}


def run_cfg4(args, sh, g, torch, np, w, h, F, r, lo, total):
    """BASELINE configs[4]: every frame goes through gs_blur(r) -> gs_sobel (zeroed dst) -> gs_integral ->
    gs_lbp_detect(frontalface, sf 1.1, scales 1..4, step 1, max_rects 4096) on the GPU that owns it; frames
    are sharded by index.  What crosses GPUs: the cascade blob broadcast from rank 0, the timing
    barrier / max, the per-frame counts and the packed variable-length gs_rect lists -- never a pixel plane."""
    from grayskull_amd.cascade import Cascade
    blob = open(os.path.join(ROOT, "tests", "golden", "frontalface_cascade.bin"), "rb").read() if sh.rank == 0 else b""
    casc = Cascade.from_bytes(sh.broadcast_bytes(blob, root=0), "broadcast from rank 0")
    G = 16  # frames per inner group (bounds the u32 integral scratch: 16 x 33 MB)
    src = torch.empty((G, h, w), dtype=torch.uint8, device="cuda")
    a, b = torch.empty_like(src), torch.empty_like(src)
    ii = torch.zeros((G, h, w), dtype=torch.int32, device="cuda")
    rects = torch.zeros((F, 4096, 4), dtype=torch.int32, device="cuda")  # 64 KB per frame: every frame's list is kept
    counts = torch.zeros(F, dtype=torch.int32, device="cuda")
    evaluated = torch.zeros(4, dtype=torch.int64, device="cuda")
    dc = g.cascade_create(casc)

    def step():
        for f0 in range(0, F, G):
            n = min(G, F - f0)
            g.synth_batch(src[:n], 1000 + lo + f0)  # frames are generated where they are processed
            g.blur_batch(a[:n], src[:n], r)
            b[:n].zero_()
            g.sobel_batch(b[:n], a[:n])
            g.integral_batch(b[:n], ii[:n])
            g.lbp_detect_batch(dc, ii[:n], rects[f0:f0 + n], counts[f0:f0 + n], 4096, 1.1, 1.0, 4.0, 1)

    for _ in range(args.warmup):
        step()
    sh.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0
    sh.barrier()
    dt = sh.max_over_ranks(time.perf_counter() - t0)
    rank_ms = rank_times(sh, torch, dt_local, args.steps)
    g.lbp_count_evaluated(evaluated)  # one more, untimed step with the counting build of the cascade kernel
    step()
    torch.cuda.synchronize()
    g.lbp_count_evaluated(None)
    all_counts, all_rects = sh.gather_varlen(counts, rects, total)  # every frame's rect list, global frame order
    ev_total = sh.sum_over_ranks(float(evaluated[0].item())) * args.steps
    nwin = g.lbp_window_count(casc, w, h, 1.1, 1.0, 4.0, 1)
    # ---- verification (outside the timed region): reference-generated golden lists, every rank's frames ----
    parity = "skipped"
    if not args.no_verify:
        gold = load_golden_batch(w, h, r)
        starts = torch.cumsum(all_counts, 0) - all_counts
        checked, bad = [], []
        if gold and sh.rank == 0:
            rc = all_rects.cpu().numpy()
            for fs, want in sorted(gold["cfg4"]["frames"].items(), key=lambda kv: int(kv[0])):
                f = int(fs)
                if f >= total:
                    continue
                n0, s0 = int(all_counts[f]), int(starts[f])
                got = {"n": n0, "wsum": "%016x" % wsum_bytes(np, rc[s0:s0 + n0])}
                checked.append(f)
                if got != want:
                    bad.append(f)
        live = None
        if sh.rank == 0 and args.verify_live:  # frame 0 against the CPU oracle itself (~30 s of one host core at 4K)
            from oracle import pyoracle
            o = pyoracle.Oracle("reference" if pyoracle.have_reference() else "port")
            e = o.sobel(o.blur(pyoracle.Oracle.synth(w, h, 1000), r))
            want = o.lbp_detect(casc, o.integral(e), 4096, 1.1, 1.0, 4.0, 1)
            n0 = int(all_counts[0])
            got = all_rects[:n0].cpu().numpy().astype(np.uint32)
            live = n0 == len(want) and np.array_equal(got.reshape(-1), want.view(np.uint32).reshape(-1))
            if not live:
                bad.append("live:0")
        if sh.rank == 0:
            parity = ("MISMATCH at frames %s" % bad if bad else
                      "rect lists of %d of the %d frames (all ranks; frames %s) == reference golden (count + checksum)%s"
                      % (len(checked), total, frame_ranges(checked), "; frame 0 bit-exact vs %s oracle live" % o.kind if live else ""))
    ranks_seen = sh.ranks_seen()  # a collective: every rank takes part (it used to sit inside rank 0's print -- found by the 8-rank rehearsal)
    if sh.rank == 0:
        print(json.dumps({
            "metric": "frames/s for gs_blur->gs_sobel->gs_integral->gs_lbp_detect on 4K uint8 (BASELINE configs[4])",
            "value": round(total * args.steps / dt, 2), "unit": "frames/s", "n_gpus": sh.world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "configs[4]: %d frames %dx%d on this GPU, %d in total, sharded by frame" % (F, w, h, total),
                       "frames_per_gpu": F, "global_frames": total},
            "Mpix/s": round(total * w * h * args.steps / dt / 1e6, 1),
            "windows_per_frame_full_scan": nwin,
            "windows_evaluated_per_frame": round(ev_total / (total * args.steps), 1),
            "Gwindows/s_evaluated": round(ev_total / dt / 1e9, 2),
            "rccl_ranks_seen": ranks_seen, "backend": sh.backend, "rank_ms_per_step": rank_ms,
            "collectives": ["broadcast(cascade blob, %d B)" % len(blob), "barrier", "all_reduce(max, sum)",
                            "all_gather(counts)", "all_gather(packed gs_rect lists, %d records)" % int(all_rects.shape[0])],
            "parity": parity,
            "detections_total": int(all_counts.sum()), "detections_first_frames": all_counts[:4].cpu().tolist()}))
    dc.close()
    sh.close()


def fused_valu_block(lpx, w, fms):
    """VALU pricing of one fused-kernel launch: wave-instructions by issue class / measured issue rate.
    lpx pixels per launch -> wave-rows = lpx / 1024 * (columns of waves cover ceil(w/1024)*1024 px per row)."""
    mix, stamp = measurement(FUSED_OPS_PER_ROW_FILE)
    if mix is None:
        return {"stale_or_missing": stamp}
    wave_rows = lpx / w * ((w + 1023) // 1024)
    sec = {k: wave_rows * mix["per_wave_row"][k] / (VALU_RATE_GINST[k] * 1e9) for k in ("full", "half")}
    t = sec["full"] + sec["half"]
    return {"wave_instructions_per_wave_row": mix["per_wave_row"], "source": mix.get("source"), "taken_at": stamp,
            "issue_rate_Gwaveinst_s": VALU_RATE_GINST, "valu_ms_at_measured_issue_rate": round(t * 1e3, 4),
            "valu_frac": round(t * 1e3 / fms, 4),
            "lds_atomics_per_wave_row": mix["per_wave_row"].get("ds_add"),
            "note": "valu_frac = time the kernel's VALU instructions need at the chip's measured issue rates / measured "
                    "launch time; rates from profiles/r02a_ubench_valu.log"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=None,
                    help="weak scaling (default): frames PER GPU, default 512 (512 x 8.3 MB = 4.25 GB/plane >> 256 MiB L3, 3 planes "
                         "resident; at --gpus 8 that is BASELINE configs[4]'s 4096-frame batch).  --scaling strong: frames of the "
                         "WHOLE job, default 4096, split over the ranks in contiguous balanced blocks (N = 1 holds all of them: "
                         "3 planes x 34 GB of the 288 GB)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: per-GPU work fixed as N grows (the driver's scaling runs).  strong: the global batch is fixed -- "
                         "north_star's '>= 7x at 8 GPUs vs 1 over a 4096-frame batch' is value(N = 8) / value(N = 1) of this mode")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--radius", type=int, default=2)
    ap.add_argument("--workload", default="cfg1", choices=["cfg1", "cfg4"],
                    help="cfg1 (default): BASELINE configs[1], the metric's workload.  cfg4: configs[4], per frame "
                         "gs_blur(2) -> gs_sobel -> gs_integral -> gs_lbp_detect, frames sharded over the GPUs")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--verify-live", action="store_true", help="cfg4: also run frame 0 through the CPU oracle (~30 s)")
    ap.add_argument("--no-other", action="store_true", help="skip the other_configs / per-kernel extras (profiling runs)")
    args = ap.parse_args()
    spawn_ranks_if_needed(args)

    import numpy as np
    import torch
    import grayskull_amd as gs
    from grayskull_amd.shard import Sharder

    # GS_BENCH_BACKEND=gloo GS_BENCH_DEVICE=0: rehearse the N>1 code path on a 1-GPU box (ranks
    # share the GPU, so the number means nothing); the driver's runs use neither.
    sh = Sharder(backend=os.environ.get("GS_BENCH_BACKEND") or None)
    if sh.world != args.gpus:
        sys.exit("bench.py: WORLD_SIZE=%d does not match --gpus %d" % (sh.world, args.gpus))
    device = int(os.environ.get("GS_BENCH_DEVICE", sh.local_rank))
    if device >= torch.cuda.device_count():
        sys.exit("bench.py: rank %d wants cuda:%d but only %d GPU(s) are visible" % (sh.rank, device, torch.cuda.device_count()))
    torch.cuda.set_device(device)
    # GS_BENCH_LIB: an experiment build of the library (build_variants/, A/B runs only)
    g = gs.Grayskull(os.environ["GS_BENCH_LIB"]) if os.environ.get("GS_BENCH_LIB") else gs.lib()
    g.set_device(device)
    g.use_torch_stream()

    w, h, r = args.width, args.height, args.radius
    if args.scaling == "strong":  # the global batch is fixed: contiguous balanced blocks (grayskull_amd/shard.py frame_range)
        from grayskull_amd.shard import frame_range
        total = 4096 if args.frames is None else args.frames
        lo, hi = frame_range(sh.rank, sh.world, total)
        F = hi - lo
    else:  # weak scaling: every rank owns F frames, global index lo..lo+F
        F = 512 if args.frames is None else args.frames
        lo, total = sh.rank * F, F * sh.world
    if total < 1:
        sys.exit("bench.py: --frames must be >= 1")
    if args.workload == "cfg4":  # ranks that own no frame (world > frames) take part in every collective with empty shares
        return run_cfg4(args, sh, g, torch, np, w, h, F, r, lo, total)
    if total < sh.world:  # the same verdict on every rank: nobody is left waiting in a collective
        sys.exit("bench.py: %d frames cannot be split over %d ranks in the configs[1] workload (per-launch figures need a "
                 "non-empty share on every rank); use --workload cfg4 to rehearse empty shares" % (total, sh.world))
    src = torch.empty((F, h, w), dtype=torch.uint8, device="cuda")
    tmp = torch.empty_like(src)
    dst = torch.empty_like(src)
    hist = torch.zeros((F, 256), dtype=torch.int32, device="cuda")
    thr = torch.zeros(F, dtype=torch.uint8, device="cuda")
    g.synth_batch(src, 1000 + lo)  # inputs resident in HBM before the timed region
    torch.cuda.synchronize()

    def step():  # tmp=None: the blurred frames are not requested -> fused blur+sobel+histogram kernel
        g.edge_pipeline_batch(dst, None, src, r, hist, thr)

    def step_unfused():  # the same chain as separate per-call kernels (blurred frames materialised)
        g.edge_pipeline_batch(dst, tmp, src, r, hist, thr)

    # the library brackets every launch of the dominant (fused) kernel with HIP events on the stream
    # it is launched on -- during the timed steps themselves; the pairs are created beforehand
    launches_per_step = (F + 31) // 32
    g.profile(max(2, (args.steps + args.warmup) * launches_per_step + 8))
    for _ in range(args.warmup):
        step()
    g.profile_read()  # drop the warm-up launches
    sh.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt_local = time.perf_counter() - t0
    sh.barrier()
    dt = sh.max_over_ranks(time.perf_counter() - t0)
    rank_ms = rank_times(sh, torch, dt_local, args.steps)
    nl, tot_ms = g.profile_read()  # launches of the timed region
    g.profile(False)
    npx = F * w * h
    value = float(total) * w * h * args.steps / dt / 1e6
    ms_step = dt / args.steps * 1e3

    # ---- per-kernel timing on the launch stream (after the timed region) -------------------
    reps = max(5, min(args.steps, 20))
    ms_unfused = time_stream(torch, step_unfused, reps)
    ms_fused = time_stream(torch, step, reps)
    bs = "gs_blur(r=%d)+gs_sobel, one pass (gsh_blur_sobel_batch)" % r
    # (call, bytes the launch really moves, SURVEY 8d per-call bytes)
    kernels = {
        bs: (lambda: g.blur_sobel_batch(dst, src, r), 2.0 * npx, 4.0 * npx),
        "gs_blur(r=%d) k_blur16" % r: (lambda: g.blur_batch(tmp, src, r), 2.0 * npx, 2.0 * npx),
        "gs_sobel k_sobel16": (lambda: g.sobel_batch(dst, tmp), float(F * (w * h + (w - 2) * (h - 2))),
                               float(F * (w * h + (w - 2) * (h - 2)))),
        "gs_histogram+otsu": (lambda: g.otsu_batch(dst, hist, thr), 1.0 * npx, 1.0 * npx),
        "gs_threshold k_threshold": (lambda: g.threshold_batch(dst, thr), 2.0 * npx, 2.0 * npx),
        "gs_erode k_morph16": (lambda: g.erode_batch(dst, src), 2.0 * npx, 2.0 * npx),
        # radii <= 16 keep the window's rows in a register ring (k_box16r)
        "gs_blur(r=5) k_box16r": (lambda: g.blur_batch(tmp, src, 5), 2.0 * npx, 2.0 * npx),
        "gs_adaptive_threshold(r=15, c=5) k_box16r": (lambda: g.adaptive_threshold_batch(tmp, src, 15, 5), 2.0 * npx, 2.0 * npx),
    }
    # gs_integral on 64 frames (u32 table: 2.1 GB): bytes moved from the PMC passes when they cover this shape
    n_ii = min(64, F)
    ii_buf = torch.empty((n_ii, h, w), dtype=torch.int32, device="cuda")
    moved_ii = 5.0 * n_ii * w * h
    csrc = "grayskull_amd/csrc/"
    pt_all, pt_stamp = measurement(os.path.join(ROOT, "profiles", "pmc_traffic.json"),
                                   only=[csrc + "k_fused.h", csrc + "gs_fused.cpp", csrc + "k_strip.h"])
    pt_ii, _ = measurement(os.path.join(ROOT, "profiles", "pmc_traffic.json"), only=[csrc + "k_integral.h"])
    try:
        if pt_ii and (w, h, n_ii) == (3840, 2160, 64):
            moved_ii = float(sum(v["total_bytes"] for k, v in pt_ii["per_launch"].items() if k.startswith("gs::k_integral")))
    except Exception:
        pass
    kernels["gs_integral (3 launches: colsum, colbase, wave), %d frames" % n_ii] = (
        lambda: g.integral_batch(src[:n_ii], ii_buf), moved_ii, 5.0 * n_ii * w * h)
    ktab = {}
    for name, (fn, moved, percall) in kernels.items():
        ms = time_stream(torch, fn, reps)
        ktab[name] = hbm_block(moved, ms)
        if percall != moved:
            ktab[name]["percall_equivalent_GB/s"] = round(percall / ms / 1e6, 1)
        if name.startswith("gs_histogram"):  # one ds_add_u32 per pixel: the other physical ceiling of this kernel
            ktab[name]["lds_atomic_frac"] = round(npx / ms / 1e9 / LDS_ATOMIC_TLANE_S, 4)
            ktab[name]["lds_atomic_note"] = ("pixels/s / %.1f T lane-atomics/s, the chip's measured ds_add_u32 rate "
                                             "(profiles/r02i_ubench_new_ops.log, pure stream, 4.1 cycles per wave per CU)" % LDS_ATOMIC_TLANE_S)
    del ii_buf
    # The fused kernel itself: average over ALL its launches inside the timed region above (HIP events
    # recorded by the library on the launch stream, gsh_profile).  A step launches it once per
    # 32-frame chunk; chunk i's threshold pass runs on a side stream under chunk i+1's fused
    # kernel, so these durations include that sharing.
    fk = "fused blur+sobel+hist k_blur_sobel_hist16"
    fms = tot_ms / max(nl, 1)
    fpl = F / launches_per_step           # frames per launch (nl may be capped at 4096 bracketed launches)
    lpx = fpl * w * h                     # pixels per launch
    ktab[fk] = hbm_block(2.0 * lpx, fms, frames_per_launch=fpl, launches_timed=nl,
                         note="per launch, HIP events on the launch stream over the timed region; 1 R + 1 W per pixel")
    pmc = None
    try:
        pt = pt_all
        # the kernel's printed name carries its template arguments ("<2>" or "<2, true>"): match by prefix
        names = [k for k in pt["per_launch"] if k.startswith("gs::k_blur_sobel_hist16<2") and "false" not in k]
        if (w, h, r) == (3840, 2160, 2) and names and pt.get("fused_frames_per_launch") == fpl:
            pmc = pt["per_launch"][names[0]].get("total_bytes")
    except Exception:
        pass
    moved = 2.0 * lpx
    roof = {"bound": "hbm", "kernel": "k_blur_sobel_hist16<2> (gs_blur r=2 + gs_sobel + gs_histogram in one launch)",
            "achieved": round(moved / fms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(moved / fms / 1e6 / HBM_PEAK_GBS, 4), "traffic": pmc,
            "accounting": "physical: bytes the launch moves (1 read + 1 write per pixel) / HIP-event launch time / 8 TB/s",
            "algorithmic_bytes_per_launch": moved, "avg_launch_ms": round(fms, 4), "frames_per_launch": fpl,
            "frac_from_pmc_traffic": round(pmc / fms / 1e6 / HBM_PEAK_GBS, 4) if pmc else None,
            "limited_by": "valu", "valu": fused_valu_block(lpx, w, fms),
            "percall_equivalent": {"bytes_per_px": 5, "GB/s": round(5.0 * lpx / fms / 1e6, 1),
                                   "note": "SURVEY 8(d): the three calls this launch replaces (gs_blur 2 + gs_sobel 2 + "
                                           "gs_histogram 1 B/px) would move 5 B/px; a fusion-equivalent throughput, "
                                           "not a roofline fraction (it can exceed the peak)"},
            "chain_physical": {"bytes_per_px": 4, "GB/s": round(4.0 * npx / ms_step / 1e6, 1),
                               "frac": round(4.0 * npx / ms_step / 1e6 / HBM_PEAK_GBS, 4),
                               "frac_of_copy_ceiling": round(4.0 * npx / ms_step / 1e6 / HBM_COPY_GBS, 4),
                               "note": "whole timed step: fused pass 1 R + 1 W, threshold pass 1 R + 1 W per pixel"},
            "per_call_kernels": {k: ktab[k]["frac"] for k in ktab if k != fk},
            "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc, corrected per MI355X_MICROARCH.md)" if pmc else None,
            "traffic_taken_at": pt_stamp}

    ns = other = None
    if sh.world == 1 and not args.no_other:
        ns, other = extras(g, torch, np, src, tmp, dst, w, h, reps, lo, args)
        # flat copies of what north_star sets a target on and of SURVEY 8(d)'s other per-path figures: the driver's record
        # keeps the scalar entries of `roofline`
        roof["north_star_kernel"] = "k_sobel16, gs_sobel alone on %d distinct 4096x4096 frames per launch" % ns["frames"]
        roof["north_star_Mpix_s"], roof["north_star_frac"], roof["north_star_ms"] = ns["Mpix/s"], ns["frac_hbm_peak"], ns["ms_per_launch"]
        roof["north_star_target_frac"] = NORTH_STAR_TARGET_FRAC  # BASELINE.json's constant, beside the measured figure
        try:
            c3 = other["configs[3] gs_orb_extract x2 + gs_match_orb 1280x720 threshold=20 nkps=500"]
            for k, v in c3["gs_match_orb_device_resident"].items():
                roof["match_orb_%s_Gpairs_s" % k], roof["match_orb_%s_ms" % k] = v["Gpairs/s"], v["ms"]
                roof["match_orb_issue_bound_Gpairs_s"] = v["issue_bound_Gpairs/s"]
            roof["gs_fast_32x720p_ms"] = c3["gs_fast_roofline"]["ms"]
            roof["orb_extract_libm_batch_us_per_frame"] = round(c3["libm_batch_orb_extract_ms_per_frame"] * 1e3, 2)
            roof["orb_extract_nostdlib_us_per_frame"] = round(c3["device_resident_orb_extract_ms_per_frame"] * 1e3, 2)
            c2 = other["configs[2] gs_integral + gs_lbp_detect(frontalface) 1920x1080 sf=1.1 scales 1..4 step 1"]
            roof["cfg2_lbp_ms_per_frame"] = c2["lbp_ms_per_frame"]
            c4 = [v for k, v in other.items() if k.startswith("configs[4]")][0]
            roof["cfg4_frames_per_s_per_gpu"], roof["cfg4_lbp_ms_per_frame"] = c4["frames_per_s_per_gpu"], c4["lbp_ms_per_frame"]
        except Exception as e:  # a missing side block must not cost the headline line
            roof["side_blocks_error"] = repr(e)

    # ---- verification (outside the timed region) ----------------------------------------------
    # (1) EVERY rank checks sample frames of its own shard bit for bit against the CPU oracle (the compiled
    #     reference when oracle/_ref is there, else the C restatement); the verdicts are AND-ed over ranks.
    # (2) every rank checksums ALL its output frames on its GPU; the per-frame checksums and Otsu thresholds are
    #     gathered in global frame order (KB-scale, RCCL when N > 1) and rank 0 compares every one of them with
    #     tests/golden/batch_checksums.json, which the unmodified reference generated for all 4096 frames.
    step()
    torch.cuda.synchronize()
    thr_all = sh.all_gather_frames(thr, total)
    sums = torch.zeros(F, dtype=torch.int64, device="cuda")
    g.checksum_batch(dst, sums)
    torch.cuda.synchronize()
    sums_all = sh.all_gather_frames(sums, total)
    sums_list = [v & 0xffffffffffffffff for v in sums_all.cpu().tolist()]
    parity, golden_note, oracle_kind = "skipped", None, None
    if not args.no_verify:
        from oracle import pyoracle
        oracle_kind = "reference" if pyoracle.have_reference() else "port"
        o = pyoracle.Oracle(oracle_kind)
        ok = True
        vf = sorted({0, min(31, F - 1), min(32, F - 1), F // 2, F - 1})  # both sides of a chunk boundary
        for f in vf:
            img = pyoracle.Oracle.synth(w, h, 1000 + lo + f)
            e = o.sobel(o.blur(img, r))
            t = o.otsu_threshold(e)
            ok &= int(thr[f]) == t and np.array_equal(dst[f].cpu().numpy(), o.threshold(e, t))
        ok_all = sh.min_over_ranks(1.0 if ok else 0.0) > 0.5
        parity = ("bit-exact vs %s oracle on local frames %s of every rank (%d ranks)" % (oracle_kind, vf, sh.world)
                  if ok_all else "MISMATCH vs oracle")
        gold = load_golden_batch(w, h, r)
        if gold and sh.rank == 0:
            nchk = min(total, gold["frames"])
            bad = [f for f in range(nchk) if "%016x" % sums_list[f] != gold["cfg1"]["wsum"][f]
                   or int(thr_all[f]) != gold["cfg1"]["otsu"][f]]
            golden_note = ("%d/%d checksums + otsu thresholds match golden (reference-generated, all ranks' frames)"
                           % (nchk - len(bad), nchk))
            if bad:
                parity = "MISMATCH vs golden at global frames %s" % bad[:16]
            else:
                parity = golden_note + "; " + parity
    # SURVEY 8(e)'s "checksum of checksums": the same number whatever N is for the same global frames
    digest = 1469598103934665603
    for v in sums_list:
        digest = ((digest ^ v) * 1099511628211) & 0xffffffffffffffff
    ranks_seen = sh.ranks_seen()

    out = {
        "metric": BASELINE_METRIC,  # BASELINE.json's metric string; the workload is named in config.workload
        "value": round(value, 1), "unit": "Mpix/s", "n_gpus": sh.world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_step, 4),
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": "configs[1]: gs_blur(r=%d) -> gs_sobel -> gs_otsu_threshold -> gs_threshold, "
                               "%dx%d uint8, %d frames on this GPU resident in HBM (%d in the job, %s scaling)" % (r, w, h, F, total, args.scaling),
                   "frames_per_gpu": F, "global_frames": total, "sharding": "by frame, contiguous balanced blocks, no data-path collective",
                   "chain_algorithmic_bytes_per_px_unfused": 7, "hbm_peak_GBs": HBM_PEAK_GBS},
        "rccl_ranks_seen": ranks_seen, "backend": sh.backend, "rank_ms_per_step": rank_ms,
        "fused": {"ms_per_step": round(ms_fused, 4), "Mpix/s": round(npx / ms_fused / 1e3, 1),
                  "hbm_bytes_per_px": 4, "hbm_GB/s": round(4.0 * npx / ms_fused / 1e6, 1),
                  "hbm_frac_of_peak": round(4.0 * npx / ms_fused / 1e6 / HBM_PEAK_GBS, 4), "note": "blur+sobel+histogram in one kernel per 32-frame chunk; each chunk's threshold pass runs under the next chunk's fused kernel"},
        "unfused": {"ms_per_step": round(ms_unfused, 4), "Mpix/s": round(npx / ms_unfused / 1e3, 1),
                    "hbm_bytes_per_px": 7, "hbm_GB/s": round(7.0 * npx / ms_unfused / 1e6, 1),
                    "hbm_frac_of_peak": round(7.0 * npx / ms_unfused / 1e6 / HBM_PEAK_GBS, 4),
                    "note": "separate gs_blur, gs_sobel, histogram, threshold kernels"},
        "blur_sobel_only": {  # BASELINE.json's metric names gs_sobel+gs_blur: that chain alone, one pass
            "ms_per_step": ktab[bs]["ms"], "Mpix/s": round(npx / ktab[bs]["ms"] / 1e3, 1),
            "hbm_GB/s (2 B/px moved)": ktab[bs]["GB/s"], "frac": ktab[bs]["frac"],
            "percall_equivalent_GB/s (4 B/px)": ktab[bs].get("percall_equivalent_GB/s"), "limited_by": "valu"},
        "roofline": roof, "kernels": ktab, "sobel_4096x4096": ns, "other_configs": other, "parity": parity,
        "otsu_thresholds_gathered": int(thr_all.numel()),
        "parity_golden": golden_note, "parity_oracle_kind": oracle_kind,
        "output_checksum_of_checksums": "%016x" % digest, "output_checksums_gathered": int(sums_all.numel()),
    }
    if sh.rank == 0 and sh.world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(w, h, r)
    if sh.rank == 0:
        print(json.dumps(out))
    sh.close()


def ragged_block(g, torch, reps):
    """the strip kernels on ragged widths (w % 16 != 0) and on frames at an odd byte address, each
    beside the same kernel on the aligned shape of the same area -- `vs_aligned` = frac / aligned frac"""
    def bufs(n, hh, ww, off=0):
        s = torch.randint(0, 256, (n * hh * ww + 64,), dtype=torch.uint8, device="cuda")
        d = torch.zeros(n * hh * ww + 64, dtype=torch.uint8, device="cuda")
        return s[off:off + n * hh * ww].view(n, hh, ww), d[off:off + n * hh * ww].view(n, hh, ww)
    out = {}

    def pair(name, fn, n, shape_a, shape_r, off_r=0, bpp=2.0):
        (wa, ha), (wr, hr) = shape_a, shape_r
        sa, da = bufs(n, ha, wa)
        ms_a = time_stream(torch, lambda: fn(da, sa), reps)
        sr, dr = bufs(n, hr, wr, off_r)
        ms_r = time_stream(torch, lambda: fn(dr, sr), reps)
        a, b = hbm_block(bpp * n * wa * ha, ms_a), hbm_block(bpp * n * wr * hr, ms_r)
        out[name] = {"frames": n, "ragged": dict(b, shape="%dx%d%s" % (wr, hr, " at base+%d" % off_r if off_r else "")),
                     "aligned": dict(a, shape="%dx%d" % (wa, ha)), "vs_aligned": round(b["frac"] / a["frac"], 3)}
    pair("gs_sobel 3838x2160", lambda d, s: g.sobel_batch(d, s), 64, (3840, 2160), (3838, 2160))
    pair("gs_blur(2) 1080x1920", lambda d, s: g.blur_batch(d, s, 2), 64, (1920, 1080), (1080, 1920))
    pair("gs_sobel 3840x2160 at base+1", lambda d, s: g.sobel_batch(d, s), 64, (3840, 2160), (3840, 2160), 1)
    pair("gs_blur(2) 3840x2160 at base+1", lambda d, s: g.blur_batch(d, s, 2), 64, (3840, 2160), (3840, 2160), 1)
    pair("gs_erode 1366x768", lambda d, s: g.erode_batch(d, s), 256, (1360, 768), (1366, 768))
    pair("gs_blur(2)+gs_sobel one pass 3838x2160", lambda d, s: g.blur_sobel_batch(d, s, 2), 64, (3840, 2160), (3838, 2160))
    # gs_integral: 612 x 816 (the reference's own fixture size, testdata/receipt.pgm) beside 608 x 816
    for name, n in (("gs_integral 612x816 x64", 64), ("gs_integral 612x816 x256", 256)):
        res = {}
        for tag, ww in (("aligned", 608), ("ragged", 612)):
            s_, _ = bufs(n, 816, ww)
            ii = torch.empty((n, 816, ww), dtype=torch.int32, device="cuda")
            ms = time_stream(torch, lambda: g.integral_batch(s_, ii), reps)
            res[tag] = dict(hbm_block(5.0 * n * ww * 816, ms), shape="%dx816" % ww)
            del ii
        out[name] = {"frames": n, "ragged": res["ragged"], "aligned": res["aligned"],
                     "vs_aligned": round(res["ragged"]["frac"] / res["aligned"]["frac"], 3)}
    s8, _ = bufs(8, 4320, 7680)
    ii8 = torch.empty((8, 4320, 7680), dtype=torch.int32, device="cuda")
    out["gs_integral 7680x4320 x8 (column chunks of 4096 px)"] = hbm_block(5.0 * 8 * 7680 * 4320, time_stream(torch, lambda: g.integral_batch(s8, ii8), reps))
    return out


def extras(g, torch, np, src, tmp, dst, w, h, reps, lo=0, args=None):
    """north-star shape and the other single-GPU configs of BASELINE.json, each with its own physical
    roofline block (not the headline metric)"""
    from grayskull_amd.cascade import Cascade
    # north-star shape: gs_sobel alone on 4096x4096, rotating over 64 distinct frames (1 GiB/plane)
    n4 = 64
    a4 = torch.empty((n4, 4096, 4096), dtype=torch.uint8, device="cuda")
    b4 = torch.zeros_like(a4)
    g.synth_batch(a4, 2)
    # 100 launches behind 10 untimed ones: the first ~20 launches of this shape after other work read up to 15 % slow
    # (profiles/r05_sobel4096.log: per-launch HIP events over 320 launches of a fresh process, clocks beside them)
    ns_reps = max(reps, 100)
    ms = time_stream(torch, lambda: g.sobel_batch(b4, a4), ns_reps, warm=10)
    by = float(n4 * (4096 * 4096 + 4094 * 4094))
    ns = {"Mpix/s": round(n4 * 4096 * 4096 / ms / 1e3, 1), "GB/s": round(by / ms / 1e6, 1),
          "frac_hbm_peak": round(by / ms / 1e6 / HBM_PEAK_GBS, 4), "frames": n4, "ms_per_launch": round(ms, 4),
          "launches_timed": ns_reps,
          # constants, not measurements: what BASELINE.json's north_star asks of this kernel (>= 60 % of 8 TB/s at 2 B/px)
          "target_from_BASELINE_json": {"frac": NORTH_STAR_TARGET_FRAC, "Mpix/s": NORTH_STAR_TARGET_FRAC * HBM_PEAK_GBS * 1e3 / 2.0}}
    del a4, b4

    other = {}
    casc = Cascade.from_blob(os.path.join(ROOT, "tests", "golden", "frontalface_cascade.bin"))
    n3, h3, w3 = 8, 1080, 1920
    s3 = torch.empty((n3, h3, w3), dtype=torch.uint8, device="cuda")
    g.synth_batch(s3, 3)
    ii3 = torch.zeros((n3, h3, w3), dtype=torch.int32, device="cuda")
    rc = torch.zeros((n3, 4096, 4), dtype=torch.int32, device="cuda")
    cn = torch.zeros(n3, dtype=torch.int32, device="cuda")
    ev = torch.zeros(4, dtype=torch.int64, device="cuda")
    dc = g.cascade_create(casc)
    ms_ii = time_stream(torch, lambda: g.integral_batch(s3, ii3), 5)
    ms_lbp = time_stream(torch, lambda: g.lbp_detect_batch(dc, ii3, rc, cn, 4096, 1.1, 1.0, 4.0, 1), 3)
    g.lbp_count_evaluated(ev)
    g.lbp_detect_batch(dc, ii3, rc, cn, 4096, 1.1, 1.0, 4.0, 1)
    torch.cuda.synchronize()
    g.lbp_count_evaluated(None)
    nwin = g.lbp_window_count(casc, w3, h3, 1.1, 1.0, 4.0, 1)
    nev, nweak, nload = int(ev[0]), int(ev[1]), int(ev[2])
    other["configs[2] gs_integral + gs_lbp_detect(frontalface) 1920x1080 sf=1.1 scales 1..4 step 1"] = {
        "frames": n3, "integral_ms_per_frame": round(ms_ii / n3, 4), "lbp_ms_per_frame": round(ms_lbp / n3, 3),
        "windows_per_frame": nwin, "windows_evaluated_per_frame": nev // n3,
        "Gwindows/s": round(nev / ms_lbp / 1e6, 2),
        "detections_frame0": int(cn[0]), "expected_frame0 (reference KAT)": 158,
        "integral_roofline": hbm_block(5.0 * n3 * h3 * w3, ms_ii, bytes_per_px=5,
                                       note="algorithmic 1 R + 4 W per px; the three-launch form moves ~6.1 (PMC)"),
        "lbp_roofline": lbp_gather_block(nweak, nload, ms_lbp),
        "reference_1core": "5.98 s/frame, 4.83 Mwin/s (BASELINE.md)"}
    dc.close()
    dA = torch.empty((1, 720, 1280), dtype=torch.uint8, device="cuda")
    g.synth_batch(dA, 4)  # frame A = synth(1280, 720, seed 4) generated on the device; B = A shifted by (+5, +3), zero fill
    dA = dA[0]
    dB = torch.zeros_like(dA)
    dB[:717, :1275] = dA[3:, 5:]
    sm = torch.zeros_like(dA)
    ka = g.orb_extract_dev(dA, sm, 500, 20)
    t1 = time.perf_counter()
    for _ in range(10):
        ka = g.orb_extract_dev(dA, sm, 500, 20)
    t_orb = (time.perf_counter() - t1) / 10
    kb = g.orb_extract_dev(dB, sm, 500, 20)
    t1 = time.perf_counter()
    for _ in range(10):
        mm = g.match_orb(ka, kb, 2500, 60.0)
    t_match = (time.perf_counter() - t1) / 10
    # gs_fast alone, batched (score pass 1 R + 1 W, NMS 1 R: 3 B/px)
    nf = 32
    f7 = torch.empty((nf, 720, 1280), dtype=torch.uint8, device="cuda")
    g.synth_batch(f7, 4)
    sm7 = torch.zeros_like(f7)
    kp7 = torch.zeros((nf, 2000, 12), dtype=torch.int32, device="cuda")
    cn7 = torch.zeros(nf, dtype=torch.int32, device="cuda")
    # calls of ~70 us: 5 repetitions behind one warm-up call read 12-15 % high (first launches after other work)
    ms_fast = time_stream(torch, lambda: g.fast_batch(f7, sm7, kp7, cn7, 2000, 20), 30, warm=3)
    ms_fast_score = time_stream(torch, lambda: g.fast_score_batch(sm7, f7, 20), 30, warm=3)
    # device-resident gs_orb_extract (GS_NO_STDLIB trig, no host round trip), same 32 frames, 500 keypoints each
    ko7 = torch.zeros((nf, 500, 12), dtype=torch.int32, device="cuda")
    ms_orb_dev = time_stream(torch, lambda: g.orb_extract_batch_nostdlib(f7, sm7, ko7, cn7, 500, 20), 5)
    # the libm-flavour batch entry (device-side selection, host libm for atan2f / sinf, two host round trips for all 32 frames)
    g.orb_extract_batch_dev(f7, sm7, 500, 20)
    t1 = time.perf_counter()
    for _ in range(5):
        g.orb_extract_batch_dev(f7, sm7, 500, 20)
    t_orb_batch = (time.perf_counter() - t1) / 5
    # gs_match_orb device-resident (pairs/s against the XOR + popcount issue bound), 500 x 500 and 2500 x 2500
    match_blocks = {}
    pair = torch.stack([dA, dB])
    smp = torch.zeros_like(pair)
    for nk in (500, 2500):
        kk = torch.zeros((2, nk, 12), dtype=torch.int32, device="cuda")
        ck = torch.zeros(2, dtype=torch.int32, device="cuda")
        g.orb_extract_batch_nostdlib(pair, smp, kk, ck, nk, 20)
        torch.cuda.synchronize()
        n1, n2 = int(ck[0]), int(ck[1])
        mt = torch.zeros((max(nk, 1), 3), dtype=torch.int32, device="cuda")
        mc = torch.zeros(1, dtype=torch.int32, device="cuda")
        ms_m = time_stream(torch, lambda: g.match_orb_dev(kk[0], n1, kk[1], n2, mt, mc, nk, 60.0), 30, warm=3)
        # per pair of descriptors: 8 v_xor_b32 (full-rate encoding) + 8 v_bcnt_u32_b32 (VOP3: the 578 G rate), profiles/r02*_ubench_valu
        bound = 64.0 / (8.0 / (VALU_RATE_GINST["full"] * 1e9) + 8.0 / (VALU_RATE_GINST["half"] * 1e9))
        match_blocks["%dx%d" % (n1, n2)] = {
            "ms": round(ms_m, 4), "pairs": n1 * n2, "Gpairs/s": round(n1 * n2 / ms_m / 1e6, 2), "matches": int(mc[0]),
            "issue_bound_Gpairs/s": round(bound / 1e9, 1), "frac_of_issue_bound": round(n1 * n2 / ms_m / 1e3 / bound, 4),
            "note": "gsh_match_orb_dev, keypoints and matches resident in HBM, no host sync inside the timed region; the bound "
                    "is 64 lanes x the chip's measured VALU issue rate over 8 XOR + 8 popcount-accumulate per pair; at these "
                    "sizes the call is three dependent launches (match, scan-free emit) of microseconds each: latency, not issue"}
    del pair, smp
    other["configs[3] gs_orb_extract x2 + gs_match_orb 1280x720 threshold=20 nkps=500"] = {
        "orb_extract_ms": round(t_orb * 1e3, 3), "keypoints": int(len(ka)), "match_ms": round(t_match * 1e3, 3),
        "matches": int(len(mm)), "expected_matches (reference KAT)": 337,
        "note": "wall time incl. the host round trips (host libm atan2f/sinf, stable sort)",
        "gs_match_orb_device_resident": match_blocks,
        "libm_batch_orb_extract_ms_per_frame": round(t_orb_batch * 1e3 / nf, 4),
        "libm_batch_note": "gsh_orb_extract_batch, %d frames per call, glibc atan2f / sinf on the host (the default flavour, bit-exact "
                           "vs the reference built without GS_NO_STDLIB), wall time incl. its two host round trips" % nf,
        "device_resident_orb_extract_ms_per_frame": round(ms_orb_dev / nf, 4),
        "device_resident_note": "gsh_orb_extract_batch_nostdlib, %d frames per call, the reference's GS_NO_STDLIB trig "
                                "(ref :70-88), no host round trip; bit-exact vs the -DGS_NO_STDLIB reference build" % nf,
        "gs_fast_roofline": hbm_block(2.0 * nf * 720 * 1280, ms_fast, bytes_per_px=2, frames=nf, limited_by="valu",
                                      score_pass_ms=round(ms_fast_score, 4),
                                      score_pass_valu=fast_valu_block(nf * 720 * 1280, ms_fast_score),
                                      note="score pass 1 R + 1 W (+ 1 bit per pixel: the bitmap of scored pixels); the sparse NMS "
                                           "pass reads the bitmap and three dwords per scored pixel (round 3: a third pass over "
                                           "every pixel, 3 B/px); the score pass (LDS-tile kernel) is VALU work on real "
                                           "candidates, the NMS / emit passes behind it are latency-bound"),
        "reference_1core": "70 ms extract, 48.6 ms match (BASELINE.md)"}
    del s3, ii3, rc, cn, f7, sm7, kp7, ko7
    # configs[4], one GPU's share AT ITS REAL SIZE: every frame this GPU holds (512 by default = 4096 / 8) goes through
    # gs_blur(2) -> gs_sobel -> gs_integral -> gs_lbp_detect on 4K, in groups of 16 frames (the u32 tables of a group: 0.5 GB).
    # One timed pass (~2.3 s); the counting build of the cascade runs on the first group only.
    F5, G5 = min(512, int(src.shape[0])), 16
    a5, b5 = tmp[:G5], dst[:G5]
    ii5 = torch.zeros((G5, h, w), dtype=torch.int32, device="cuda")
    rc5 = torch.zeros((F5, 4096, 4), dtype=torch.int32, device="cuda")
    cn5 = torch.zeros(F5, dtype=torch.int32, device="cuda")
    dc5 = g.cascade_create(casc)

    def group5(f0, n, lbp=True):
        g.blur_batch(a5[:n], src[f0:f0 + n], 2)
        b5[:n].zero_()
        g.sobel_batch(b5[:n], a5[:n])
        g.integral_batch(b5[:n], ii5[:n])
        if lbp:
            g.lbp_detect_batch(dc5, ii5[:n], rc5[f0:f0 + n], cn5[f0:f0 + n], 4096, 1.1, 1.0, 4.0, 1)

    def chain5():
        for f0 in range(0, F5, G5):
            group5(f0, min(G5, F5 - f0))
    group5(0, min(G5, F5))  # warm-up
    ms5 = time_stream(torch, chain5, 1) if F5 > G5 else time_stream(torch, chain5, 2)
    n1 = min(G5, F5)
    group5(0, n1, lbp=False)
    ms5_lbp_plain = time_stream(torch, lambda: g.lbp_detect_batch(dc5, ii5[:n1], rc5[:n1], cn5[:n1], 4096, 1.1, 1.0, 4.0, 1), 2)
    ev.zero_()
    g.lbp_count_evaluated(ev)  # one untimed run of the counting build of the kernels
    g.lbp_detect_batch(dc5, ii5[:n1], rc5[:n1], cn5[:n1], 4096, 1.1, 1.0, 4.0, 1)
    torch.cuda.synchronize()
    g.lbp_count_evaluated(None)
    nev5, nweak5, nload5 = int(ev[0]), int(ev[1]), int(ev[2])
    nwin5 = g.lbp_window_count(casc, w, h, 1.1, 1.0, 4.0, 1)
    # every frame of this share that the golden file knows (reference-generated count + checksum of the rect list)
    gold5, checked5, bad5 = load_golden_batch(w, h, 2), [], []
    if gold5 and not (args and args.no_verify):
        cnh, rch = cn5.cpu().numpy(), None
        for fs, want in sorted(gold5["cfg4"]["frames"].items(), key=lambda kv: int(kv[0])):
            f = int(fs) - lo
            if 0 <= f < F5:
                rch = rc5.cpu().numpy() if rch is None else rch
                got = {"n": int(cnh[f]), "wsum": "%016x" % wsum_bytes(np, rch[f, :int(cnh[f])])}
                checked5.append(int(fs))
                if got != want:
                    bad5.append(int(fs))
    other["configs[4] per-GPU share: gs_blur(2) -> gs_sobel -> gs_integral -> gs_lbp_detect per 3840x2160 frame"] = {
        "frames": F5, "ms_per_step": round(ms5, 1), "ms_per_frame": round(ms5 / F5, 3), "frames_per_s_per_gpu": round(F5 / ms5 * 1e3, 1),
        "windows_per_frame_full_scan": nwin5, "windows_evaluated_per_frame": nev5 // n1,
        "Gwindows/s_evaluated": round(nev5 / ms5_lbp_plain / 1e6, 2), "lbp_ms_per_frame": round(ms5_lbp_plain / n1, 3), "detections": cn5.cpu().tolist()[:4],
        "lbp_roofline": lbp_gather_block(nweak5, nload5, ms5_lbp_plain),
        "parity": ("MISMATCH at global frames %s" % bad5 if bad5 else
                   "rect lists of %d of this share's %d frames (global frames %s) == reference golden (count + checksum)"
                   % (len(checked5), F5, frame_ranges(checked5))) if checked5 else "no golden frame in this share",
        "note": "one timed pass over all %d frames of this GPU in groups of %d; frames shard across GPUs with no exchange: 4096 frames on "
                "8 GPUs = 512 per GPU; the cascade dominates (120 M windows per frame; chunks behind the 4096th detection are skipped "
                "like the reference stops there)" % (F5, G5)}
    del ii5, rc5
    other["ragged widths / odd addresses (strip kernels, round 4)"] = ragged_block(g, torch, reps)
    dc5.close()
    return ns, other


LDS_B32_GBS = 128.0 * 256 * 2.4  # ds_read_b32: 128 B per clock and CU (MI355X_MICROARCH.md, LDS table), 256 CUs at 2.4 GHz


def lbp_gather_block(weak_evals, dword_loads, ms):
    """Since round 5 the cascade's corner gathers come from an LDS tile (k_lbp_tile; k_lbp_cascade on the texture path only
    for the scales whose tile does not fit).  achieved = the table dwords the lanes really loaded (gsh_lbp_count_evaluated[2])
    x 4 B / time against the ds_read_b32 rate of the chip.  The kernel is not bound by that pipe alone: on the configs[4]
    input the LDS is 75 % busy -- 37 % of its cycles are bank conflicts of the scattered survivor gathers -- and the VALU
    60-90 % (profiles/r05c_lbp_counters_tile_rule_v3.txt), i.e. both on-chip pipes sit near their limit."""
    gbs = dword_loads * 4.0 / ms / 1e6
    return {"bound": "LDS pipe (ds_read_b32 gathers from the block's table tile) + VALU issue", "weak_classifier_evaluations": weak_evals,
            "dword_loads": dword_loads, "bytes": dword_loads * 4.0, "ms": round(ms, 4), "GB/s": round(gbs, 1),
            "peak_GB/s": round(LDS_B32_GBS, 1), "frac": round(gbs / LDS_B32_GBS, 4),
            # NOT measured in this run: counters of an earlier rocprofv3 --pmc pass over the same kernels, quoted with their source
            "reference_pmc_from_profiles": {"file": "profiles/r05z_lbp_counters_rule.txt", "commit": "9419c99 2026-09-26",
                                            "lds_pipe_busy": 0.75, "lds_bank_conflict_share": 0.37,
                                            "counters": "SQ_LDS_IDX_ACTIVE / (4 x SQ_BUSY_CYCLES), SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, "
                                                        "8 x 4K edge maps"},
            "note": "bytes = table dwords loaded by the lanes (lane-level count of the counting build) x 4; peak = 128 B/clk/CU x "
                    "256 CUs x 2.4 GHz (conflict-free ds_read_b32); 16 dwords per evaluated weak classifier"}


if __name__ == "__main__":
    main()
