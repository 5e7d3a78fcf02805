#!/usr/bin/env python3
"""bench.py -- headline benchmark of the SAD disparity hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--frames-per-gpu F] [--weak]
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
          --master-port P bench.py --gpus N --steps K --warmup W)

Workload (BASELINE.json configs[2]/[3]/[4]): 1920x1080 synthetic grayscale pairs, block 9, Dmax 64.
A "step" = one pass of the hot path over this rank's shard of config 4's stream of 1024 independent
pairs: frame i -> GPU i mod N, so a rank holds 1024 / N pairs (N = 1: all 1024; "scaling": "strong" --
BASELINE config 4 is 1024 pairs IN TOTAL over 1/2/4/8 GPUs).  --weak keeps 1024 pairs per GPU instead;
--frames-per-gpu sets the shard length directly.  The shard is device-resident (generated in HBM before
the timed region), one batched launch on torch's current stream through the C ABI.  No data-path
collective; the two parameters are broadcast from rank 0 over RCCL.

Outside the timed headline, at N = 1: "host_fed" (sadhip_disparity_batch_u8 on frames in host memory,
page-locked and pageable -- what a Go / C++ host calls; PCIe-inclusive, never `value`) and "pool"
(sadhip_pool_batch_u8 over G = 1, 2, 4, 8 of the visible devices), and the roofline's HBM traffic / VALU instruction count measured in
the run itself by three child passes under rocprofv3 --pmc (--no-live-counters: the committed profile instead).

Prints ONE JSON line on rank 0: metric = stereo pairs/s (whole job), plus ms/frame, the HBM
roofline object for the disparity kernel, candidate evaluations/s, and the CPU baseline
(oracle "port" of pkg/despair timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (REPO, os.path.join(REPO, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

W, H, BLOCK, MAXDISP = 1920, 1080, 9, 64
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SIMDS, CLOCK_GHZ = 256 * 4, 2.4  # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 2.4 GHz peak engine clock
VECTOR_PEAK_GWINST = SIMDS * CLOCK_GHZ / 2     # one wave-instruction per 2 cycles per SIMD (157.3 TF fp32 = this x 64 lanes x 2)


def _source_digest():
    """sha256 over the KERNEL sources (device code, planner, launch table, generic kernels -- not sadhip.hip, the
    C-ABI host plumbing, which changes no kernel): what profiles/*.json were measured on (tools/valu_budget.py)."""
    import hashlib
    csrc = os.path.join(REPO, "steroscopic-hardware_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h", ".inc")) and f != "sadhip.hip":
            h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _usable_cpus():
    """Cores this process may really use: the affinity mask, cut down to the cgroup CPU quota when one
    is set (a GPU box hands each tenant a share of its host cores)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                n = max(1, min(n, int(float(quota) / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def _go_toolchain():
    """BASELINE.md 3.1 / SURVEY 8(d)(1): with a Go toolchain on the box the CPU baseline would be the reference itself
    (a loop like examples/run.stream.go:46-62 over pkg/despair).  Probed, never assumed: the version string, or None."""
    import shutil
    import subprocess
    go = shutil.which("go")
    if not go:
        return None
    try:
        return subprocess.run([go, "version"], capture_output=True, text=True, timeout=20).stdout.strip() or None
    except (OSError, subprocess.SubprocessError):
        return None


def _live_counters(frames, timeout_s=240):
    """HBM traffic and VALU instruction count of the headline launch, measured IN THIS RUN: three child processes
    `rocprofv3 --pmc <one counter set> -- python3 bench.py --headline-only` (separate passes, counters sampled on the
    disparity kernel only, the program directly after `--`), each launching the same `frames`-pair batch.  Returns
    {"FETCH_SIZE": KB, "WRITE_SIZE": KB, "SQ_INSTS_VALU": n, ...} per launch (mean over the child's launches), or
    {"error": ...}.  MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE in KB, gfx950 FETCH_SIZE x2."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return {"error": "rocprofv3 not found"}
    # this process is itself running under a profiler (its tool library is preloaded / named in the environment): a nested
    # rocprofv3 would inherit that -- leave the counters to the outer run and report the committed profile
    under = [k for k in ("HSA_TOOLS_LIB", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD", "ROCPROF_OUTPUT_PATH") if os.environ.get(k)]
    if under or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return {"error": "bench.py is running under a profiler (%s): no nested counter passes" % (", ".join(under) or "LD_PRELOAD")}
    got, t_all = {}, time.perf_counter()
    for counters in (["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_INSTS_VALU"]):
        td = tempfile.mkdtemp(prefix="sadhip_pmc_", dir="/tmp")
        try:
            cmd = [prof, "--kernel-trace", "--kernel-include-regex", "k_colwave|k_disparity", "--pmc", "GRBM_GUI_ACTIVE", *counters,
                   "--output-format", "csv", "-d", td, "-o", "pmc", "--",
                   sys.executable, os.path.join(REPO, "bench.py"), "--headline-only", "--steps", "2", "--warmup", "1",
                   "--frames-per-gpu", str(frames)]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout_s)
            files = glob.glob(os.path.join(td, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return {"error": f"rocprofv3 --pmc {' '.join(counters)}: exit {r.returncode}, {len(files)} csv; " + (r.stderr or r.stdout)[-300:]}
            per = {}
            for row in csv.DictReader(open(files[0])):
                if "k_colwave" in row["Kernel_Name"] or "k_disparity" in row["Kernel_Name"]:
                    d = per.setdefault(row["Dispatch_Id"], {})
                    d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            if not per:
                return {"error": "no disparity kernel dispatch in the counter csv"}
            for c in counters + ["GRBM_GUI_ACTIVE"]:
                vals = [d[c] for d in per.values() if c in d]
                if vals:
                    got[c] = sum(vals) / len(vals)
            got["launches_sampled"] = len(per)
        except (OSError, subprocess.SubprocessError, KeyError, ValueError) as e:
            return {"error": repr(e)[:300]}
        finally:
            shutil.rmtree(td, ignore_errors=True)
    got["seconds"] = time.perf_counter() - t_all
    return got


def _time_reference_c(ctx, sha):
    import subprocess
    import tempfile
    import numpy as np
    exe = os.path.join(REPO, "oracle", "_ref", "sad_ref")
    fx = os.path.join(REPO, "tests", "golden", "hardware_fixtures.npz")
    if not (os.path.exists(exe) and os.path.exists(fx)):
        return None
    d = np.load(fx)
    L, R, exp = d["patch_L"][1], d["patch_R"][1], d["exp_disp_sadc"]
    try:
        with tempfile.TemporaryDirectory() as td:
            os.mkdir(os.path.join(td, "mems"))
            L.tofile(os.path.join(td, "mems", "img_L_patch_1.raw")); R.tofile(os.path.join(td, "mems", "img_R_patch_1.raw"))
            t0 = time.perf_counter()
            subprocess.check_call([exe], cwd=td, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=60)
            cpu_s = time.perf_counter() - t0
            got = np.array([int(x, 16) for x in open(os.path.join(td, "exp_disp.mem")).read().split()], np.uint8).reshape(128, 128)
        gpu = ctx.disparity(L, R, 15, 64, sha.capi.FLAG_MIRROR_SADC)          # warm-up + check
        t0 = time.perf_counter()
        for _ in range(20):
            ctx.disparity(L, R, 15, 64, sha.capi.FLAG_MIRROR_SADC)
        gpu_s = (time.perf_counter() - t0) / 20
        return {"workload": "hardware/sad.c unmodified (oracle/_ref/sad_ref): 128x128 patch 1, WIN 15, D 64, incl. its file I/O, 1 core",
                "cpu_seconds": cpu_s, "cpu_matches_committed_fixture": bool(np.array_equal(got, exp)),
                "gpu_seconds_host_fed": gpu_s, "gpu_matches": bool(np.array_equal(gpu, exp)),
                "note": "same semantics on the GPU through SADHIP_FLAG_MIRROR_SADC (host buffers, synchronous call)"}
    except Exception as e:                                        # never fail the bench over the side measurement
        return {"error": repr(e)}


def _to_host(t):
    """Device tensor -> numpy through torch's page-locked allocator (never a device copy into pageable memory: DESIGN.md 6b)."""
    import torch
    p = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    p.copy_(t)
    torch.cuda.synchronize(t.device)
    return p.numpy().copy()


def _host_legs(sha, ctx, local, Ls, Rs, out, block, maxdisp, flags, nf):
    """SURVEY 8(d) "separately: end-to-end incl. H2D + D2H" and 8(e) "(ii) host-fed": the same frames from HOST
    memory through the entry points a Go / C++ host calls (examples/run.stream.go:46-62 prints the same per-frame
    elapsed).  Bus-inclusive; reported beside the headline, never as it."""
    import numpy as np
    capi = sha.capi
    frame = W * H
    ref = _to_host(out[:nf])                                  # the device-resident results of the timed launches
    legs = {}
    # the thread that allocates (first-touches) the page-locked frames and makes the calls runs on the CPUs next to the
    # device (sadhip_device_describe): on a two-socket box whose GPU hangs on the other node every copy would cross the
    # socket link otherwise
    place, old_aff = None, None
    try:
        place = capi.device_describe(local)
        if place["cpus"]:
            old_aff = os.sched_getaffinity(0)
            os.sched_setaffinity(0, place["cpus"])
    except (OSError, capi.SadHipError, AttributeError, KeyError, ValueError):
        old_aff = None
    try:
        legs = _host_legs_pinned(sha, ctx, local, Ls, Rs, out, block, maxdisp, flags, nf, ref)
    finally:
        if old_aff is not None:
            os.sched_setaffinity(0, old_aff)
    if place is not None:
        legs["host_placement"] = {"pci": place["pci"], "numa_node": place["numa_node"], "host_thread_pinned_to_device_cpus": old_aff is not None,
                                  "cpus": len(place["cpus"])}
    return legs


def _host_legs_pinned(sha, ctx, local, Ls, Rs, out, block, maxdisp, flags, nf, ref):
    import numpy as np
    capi = sha.capi
    frame = W * H
    legs = {}
    with capi.HostBuffer(3 * nf * frame) as hb:
        blk = hb.view((3, nf, H, W))
        blk[0] = _to_host(Ls[:nf]); blk[1] = _to_host(Rs[:nf])
        lefts, rights, outs = [blk[0, i] for i in range(nf)], [blk[1, i] for i in range(nf)], [blk[2, i] for i in range(nf)]

        def timed(fn, reps=3):
            fn()                                                  # warm-up (allocates the group buffers)
            best = None
            for _ in range(reps):
                blk[2] = 0
                t0 = time.perf_counter()
                fn()
                dt = time.perf_counter() - t0
                best = dt if best is None or dt < best else best
            return best, bool(np.array_equal(blk[2], ref))

        dt, ok = timed(lambda: ctx.disparity_batch(lefts, rights, block, maxdisp, flags, outs=outs))
        k_ms, t_ms = ctx.last_timing()
        legs["host_fed"] = {
            "value": nf / dt, "unit": "pairs/s", "frames": nf, "ms_per_frame": dt / nf * 1e3,
            "pcie_gb_per_s": 3 * frame * nf / dt / 1e9, "pcie_h2d_gb_per_s": 2 * frame * nf / dt / 1e9,
            "entry": "sadhip_disparity_batch_u8", "memory": "page-locked (sadhip_host_alloc), one contiguous block per plane",
            "kernel_ms_per_frame": k_ms / nf, "matches_device_resident": ok,
            "note": "groups of 8 frames per launch, one stream per direction (uploads / kernels / downloads chained by events), "
                    "asynchronous linear copies; bound by the bus's host-to-device direction (4.15 MB per pair in, 2.07 MB out; "
                    "tools/ubench/pcie_bw.hip: 55-57 GB/s one way, 47 + 47 with both directions busy)"}
        # single synchronous call per pair from page-locked memory: the WebUI's shape (one pair per processDepthMap)
        t0 = time.perf_counter()
        for i in range(min(nf, 32)):
            ctx.disparity(lefts[i], rights[i], block, maxdisp, flags, out=outs[i])
        one = (time.perf_counter() - t0) / min(nf, 32)
        k1, _ = ctx.last_timing()
        legs["host_fed"]["single_call_ms"] = {"page_locked": one * 1e3, "kernel": k1}
        # the same batch from pageable memory (Go heap memory is pageable: image.Gray.Pix)
        npg = min(nf, 64)
        pl = [np.array(blk[0, i]) for i in range(npg)]; pr = [np.array(blk[1, i]) for i in range(npg)]
        po = [np.zeros((H, W), np.uint8) for _ in range(npg)]          # touched: no page faults inside the timed call
        ctx.disparity_batch(pl, pr, block, maxdisp, flags, outs=po)          # warm-up over the same pages
        t0 = time.perf_counter()
        ctx.disparity_batch(pl, pr, block, maxdisp, flags, outs=po)
        dtp = time.perf_counter() - t0
        okp = all(np.array_equal(po[i], ref[i]) for i in range(npg))
        t0 = time.perf_counter()
        for i in range(min(npg, 16)):
            ctx.disparity(pl[i], pr[i], block, maxdisp, flags, out=po[i])
        onep = (time.perf_counter() - t0) / min(npg, 16)
        legs["host_fed"]["pageable"] = {"value": npg / dtp, "unit": "pairs/s", "frames": npg, "pcie_gb_per_s": 3 * frame * npg / dtp / 1e9,
                                        "matches_device_resident": okp,
                                        "note": "the host thread stages every frame through the context's page-locked planes with its copy threads, three frames in flight (the cgo shim's case)"}
        legs["host_fed"]["single_call_ms"]["pageable"] = onep * 1e3
        # ... and the call patch 0003 makes for OutputCamera (pkg/camera/output.go:134-190): two DECODED pictures in pageable memory
        # through sadhip_disparity_rgba_u8 -- gray conversion on the device (the default: 16.6 MB over the bus) and in the copy
        # threads while they stage (SADHIP_FLAG_HOST_GRAY, the shims' SetHostGray(true): 4.1 MB over the bus), same map
        rng = np.random.default_rng(11)
        pics = rng.integers(0, 256, (2, 2, H, W, 4), dtype=np.uint8); pics[..., 3] = 255      # two different opaque RGBA pairs
        maps = {}
        for name, fl in (("rgba_pageable_device_gray", flags), ("rgba_pageable_host_gray", flags | capi.FLAG_HOST_GRAY)):
            for k in (0, 1):
                maps[name] = ctx.disparity_rgba(pics[k][0], pics[k][1], block, maxdisp, fl)      # warm-up; k = 1's map is kept
            t0 = time.perf_counter()
            for r in range(16):
                ctx.disparity_rgba(pics[r & 1][0], pics[r & 1][1], block, maxdisp, fl)
            legs["host_fed"]["single_call_ms"][name] = (time.perf_counter() - t0) / 16 * 1e3
        legs["host_fed"]["single_call_ms"]["rgba_maps_identical"] = bool(np.array_equal(maps["rgba_pageable_device_gray"], maps["rgba_pageable_host_gray"]))
        # ... and the reference's OWN frames: testdata/ holds 640x480 pictures and params.go:13-18 starts the UI at 16 / 64 -- one
        # synchronous call per pair from pageable memory at that size (gray pair; RGBA pair converted on the device / in the copy threads)
        sw, sh_, sb, sd = 640, 480, 16, 64
        with capi.Context(local, sw, sh_, 1) as small:
            sp = [sha.synth.synth_pair(4000 + k, sw, sh_) for k in range(4)]
            spic = rng.integers(0, 256, (2, 2, sh_, sw, 4), dtype=np.uint8); spic[..., 3] = 255
            webui = {"frame": f"{sw}x{sh_}", "block": sb, "maxdisp": sd}
            for name, fn in (("gray_pageable", lambda k: small.disparity(sp[k & 3][0], sp[k & 3][1], sb, sd)),
                             ("rgba_pageable_device_gray", lambda k: small.disparity_rgba(spic[k & 1][0], spic[k & 1][1], sb, sd)),
                             ("rgba_pageable_host_gray", lambda k: small.disparity_rgba(spic[k & 1][0], spic[k & 1][1], sb, sd, capi.FLAG_HOST_GRAY))):
                for k in range(4):
                    fn(k)
                t0 = time.perf_counter()
                for k in range(32):
                    fn(k)
                webui[name] = (time.perf_counter() - t0) / 32 * 1e3
            webui["kernel"] = small.last_timing()[0]
        legs["host_fed"]["single_call_ms"]["reference_frame_size_640x480_defaults"] = webui
        # the native pool (frame i -> device i mod G), same page-locked frames, for G = 1, 2, 4, 8 of the visible devices:
        # SURVEY 8(e) "(ii) host-fed" -- the curve that is expected to flatten at the host's PCIe / memory limit
        ndev = capi.device_count()
        legs["pool"] = []
        for g in [g for g in (1, 2, 4, 8) if g <= max(ndev, 1)]:
            devices = [local] if g == 1 else list(range(g))
            with capi.Pool(devices, W, H) as pool:
                # entry e's frames (e, e + G, ...) live in memory its own worker thread allocated and first-touched
                # (sadhip_pool_host_alloc): every device reads and writes memory of its own NUMA node
                per = [(nf - e + g - 1) // g for e in range(g)]
                bufs = [pool.host_alloc(e, 3 * per[e] * frame) for e in range(g)]
                try:
                    views = [bufs[e].view((3, per[e], H, W)) for e in range(g)]
                    for i in range(nf):
                        views[i % g][0, i // g] = blk[0, i]; views[i % g][1, i // g] = blk[1, i]
                    pl = [views[i % g][0, i // g] for i in range(nf)]; pr = [views[i % g][1, i // g] for i in range(nf)]
                    po = [views[i % g][2, i // g] for i in range(nf)]

                    def run_pool():
                        pool.batch(pl, pr, block, maxdisp, flags, outs=po)
                    run_pool()
                    best = None
                    for _ in range(3):
                        for v in views:
                            v[2] = 0
                        t0 = time.perf_counter()
                        run_pool()
                        dtp = time.perf_counter() - t0
                        best = dtp if best is None or dtp < best else best
                    dt, ok = best, all(np.array_equal(po[i], ref[i]) for i in range(nf))
                    k_sum, wall = pool.last_timing()
                    placement = pool.describe()
                finally:
                    del views, pl, pr, po
                    for b in bufs:
                        b.close()
            legs["pool"].append({"value": nf / dt, "unit": "pairs/s", "frames": nf, "devices": devices, "pcie_gb_per_s": 3 * frame * nf / dt / 1e9,
                                 "entry": "sadhip_pool_batch_u8", "memory": "page-locked, per entry from sadhip_pool_host_alloc (allocated and first-touched by the entry's worker thread)",
                                 "kernel_ms_sum": k_sum, "matches_device_resident": ok, "workers": placement})
        legs["pool_note"] = ("one context + host thread per device, frame i -> device i mod G, each worker thread pinned to the CPUs of its "
                             "device's NUMA node (workers[].cpus; best effort), each entry's frames in page-locked memory that worker allocated "
                             "and first-touched after selecting its own device (the runtime places page-locked memory by the allocating thread's current device): meant to keep every "
                             "device's reads and writes off the socket link; where the pages land is checked against the device's NUMA node on the one-GPU box "
                             "(test_pool_workers_are_pinned_next_to_their_device), not yet on a host with devices on both sockets")
    return legs


def _emit(res):
    """Prints the ONE JSON line and returns the process's exit code: 3 when the line carries a parity_error (the kernel kind that ran
    is not the planned one, a GPU frame differs from the CPU oracle, a host-fed result differs from the device-resident one) --
    a wrong result fails the RUN the driver records, not just a JSON key -- else 0."""
    print(json.dumps(res), flush=True)
    if "parity_error" in res:
        print(f"bench.py: parity_error: {res['parity_error']}", file=sys.stderr)
        return 3
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames-per-gpu", type=int, default=0, help="pairs per GPU per step (default: 1024 / N, config 4's split; N = 1: 1024)")
    ap.add_argument("--weak", action="store_true", help="weak scaling: 1024 pairs per GPU whatever N is")
    ap.add_argument("--host-frames", type=int, default=256, help="frames of the host-fed / pool legs (N = 1 only; 0 = skip them)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-counters", action="store_true",
                    help="do not run the three rocprofv3 --pmc child passes (N = 1): roofline.traffic then comes from the committed "
                         "profiles/traffic.json, labelled as such")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the warm-up and timed launches of the headline workload (no latency / 128-pair / host-fed / CPU legs): "
                         "under rocprofv3 --kernel-trace --stats the kernel's average duration is then the headline launch's")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the multi-threaded CPU baseline (default: every core this process may use)")
    ap.add_argument("--flags", type=int, default=0, help="sadhip flags (2 = force generic kernel)")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="multi-rank rehearsal on a single GPU: every rank uses cuda:0 and the collectives run "
                         "over gloo on CPU tensors (the real N-GPU run uses nccl = RCCL, one GPU per rank)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import steroscopic_hardware_amd as sha

    local = 0 if args.rehearse_one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = torch.device("cpu") if args.rehearse_one_gpu else dev          # where collectives run
    dist, rank, world, _ = sha.shard.init("gloo" if args.rehearse_one_gpu else "nccl", dev)
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)

    # parameters: rank 0 decides, everyone else receives them (2 x int32 over RCCL/xGMI)
    block, maxdisp = sha.shard.broadcast_params(dist, rank, BLOCK, MAXDISP, cdev)

    STREAM = 1024                                                 # BASELINE config 4: 1024 independent pairs
    if args.frames_per_gpu > 0:
        F, scaling = args.frames_per_gpu, "weak"                  # a fixed shard per GPU
    elif args.weak:
        F, scaling = STREAM, "weak"
    else:
        F, scaling = (STREAM + world - 1) // world, "strong"      # config 4: the stream split over the GPUs (N = 1: all of it)
    seeds = sha.shard.frame_seeds(rank, world, F)                 # frame i -> GPU i mod G
    Ls, Rs = [], []
    for c in range(0, F, 16):                                     # generate in HBM, in chunks
        l, r = sha.synth.synth_batch_torch(seeds[c:c + 16], W, H, dev)
        Ls.append(l); Rs.append(r)
    Ls = torch.cat(Ls); Rs = torch.cat(Rs)
    out = torch.empty_like(Ls)
    ctx = sha.capi.Context(local, W, H, 1)
    stream = torch.cuda.current_stream(dev)

    def step():
        ctx.disparity_dev(Ls, Rs, out, block, maxdisp, args.flags, stream.cuda_stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    sha.shard.barrier(dist)
    torch.cuda.synchronize(dev)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for e0, e1 in evs:                                            # HIP events on the launch stream
        e0.record(stream)
        step()
        e1.record(stream)
    torch.cuda.synchronize(dev)
    sha.shard.barrier(dist)
    torch.cuda.synchronize(dev)
    elapsed = sha.shard.max_over_ranks(dist, time.perf_counter() - t0, cdev)
    launch_ms = [e0.elapsed_time(e1) for e0, e1 in evs]
    avg_launch_ms = sum(launch_ms) / len(launch_ms)

    if args.headline_only:
        args.no_cpu_baseline, args.host_frames, args.no_live_counters = True, 0, True
    # single-pair latency (ms/frame of one 1080p pair on an otherwise idle GPU)
    one_ms = []
    for _ in range(0 if args.headline_only else 20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        ctx.disparity_dev(Ls[:1], Rs[:1], out[:1], block, maxdisp, args.flags, stream.cuda_stream)
        e1.record(stream)
        torch.cuda.synchronize(dev)
        one_ms.append(e0.elapsed_time(e1))
    one_ms = sorted(one_ms)[len(one_ms) // 2] if one_ms else None
    # ... and per pair at 128 pairs per launch (config 4's per-GPU share on 8 GPUs; what rounds 1 and 2a measured)
    sub_ms = None
    if F > 128 and not args.headline_only:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ctx.disparity_dev(Ls[:128], Rs[:128], out[:128], block, maxdisp, args.flags, stream.cuda_stream)
        e0.record(stream)
        for _ in range(10):
            ctx.disparity_dev(Ls[:128], Rs[:128], out[:128], block, maxdisp, args.flags, stream.cuda_stream)
        e1.record(stream)
        torch.cuda.synchronize(dev)
        sub_ms = e0.elapsed_time(e1) / 10 / 128

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    total_frames = F * world * args.steps
    alg_bytes_per_launch = 3 * W * H * F                          # read 2WH + write WH per pair
    cands = sha.synth.candidates_per_pair(W, H, block, maxdisp)
    achieved_gbps = alg_bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9
    res = {
        "metric": "stereo pairs/sec, 1920x1080 gray, block=9 Dmax=64",
        "value": total_frames / elapsed,
        "unit": "pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "distributed": {"backend": (dist.get_backend() if dist is not None else None), "world_size": (dist.get_world_size() if dist is not None else 1),
                        "collectives": "2 x int32 parameter broadcast, one barrier either side of the timed region, max-over-ranks of the time; none on the data path"},
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[2]/[3]: {W}x{H} synthetic pairs (splitmix64, seeds 1000+i), "
                               f"block={block}, Dmax={maxdisp}; {F} device-resident pairs per GPU per step"
                               + (f" = BASELINE config 4's stream of {STREAM} pairs split over {world} GPU(s)" if scaling == "strong" else "")
                               + ", frames sharded round-robin, params broadcast only",
                   "frames_per_gpu_per_step": F, "width": W, "height": H, "block": block, "maxdisp": maxdisp,
                   "kernel_kind": sha.capi.kernel_kind(W, H, block, maxdisp, args.flags)},
        "ms_per_frame": avg_launch_ms / F,
        "ms_per_frame_single_pair_latency": one_ms,
        "ms_per_frame_at_128_pairs_per_launch": sub_ms,
        "gcand_per_s": cands * F / (avg_launch_ms * 1e-3) / 1e9,
        "roofline": {"bound": "hbm", "achieved": achieved_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved_gbps / HBM_PEAK_GBPS, "traffic": None,
                     "kernel_avg_launch_ms": avg_launch_ms, "algorithmic_bytes_per_launch": alg_bytes_per_launch,
                     "note": "path is integer-VALU bound (SURVEY 8d); see gcand_per_s and DESIGN.md. At BlockSize 9 the kernel holds "
                             "two forms of the row loop (triple-column for the tiles left of the image's last two columns, plain for "
                             "the last tile) and picks per workgroup"},
    }

    ran = sha.capi.last_kernel_kind()                             # what the timed launches really ran (no silent fallback)
    res["config"]["kernel_kind_ran"] = ran
    if ran != res["config"]["kernel_kind"]:
        res["parity_error"] = f"planned kernel kind {res['config']['kernel_kind']} but kind {ran} ran"

    # Counter-derived figures come from the committed rocprofv3 profile of THIS launch shape and THESE
    # kernel sources (profiles/traffic.json, profiles/valu_budget.json carry the digest of csrc/); a
    # profile of other sources is reported as stale, never as current.
    digest = _source_digest()
    same_shape, scale = False, 1.0
    try:
        tj = json.load(open(os.path.join(REPO, "profiles", "traffic.json")))
        # counters measured at THIS launch shape (same frames per launch) are reported as measured; a profile taken at
        # another batch length is scaled per pair and LABELLED extrapolated (round 2 had to: its counter passes died at
        # 1024 pairs -- tools/profile_cfg.sh now samples the disparity kernels only and runs at the bench's own shape)
        same_shape = (tj["width"], tj["height"], tj["block"], tj["maxdisp"]) == (W, H, block, maxdisp) and \
            (tj["frames_per_launch"] == F or min(tj["frames_per_launch"], F) >= 48)
        scale = F / tj["frames_per_launch"]
        if same_shape and res["config"]["kernel_kind"] == 1:
            if tj.get("source_digest") == digest:
                res["roofline"]["traffic"] = int(tj["hbm_bytes_per_launch"] * scale)
                res["roofline"]["traffic_source"] = tj.get("source", "profiles/traffic.json") + \
                    ("" if scale == 1 else " -- EXTRAPOLATED: measured at %d pairs per launch, scaled per pair to %d" % (tj["frames_per_launch"], F))
                res["roofline"]["traffic_scale"] = scale
            else:
                res["roofline"]["traffic_source"] = "stale: profiles/traffic.json was measured on other kernel sources (digest %s, now %s)" % (tj.get("source_digest"), digest)
    except (OSError, KeyError, ValueError):
        tj = None
    try:
        vb = json.load(open(os.path.join(REPO, "profiles", "valu_budget.json")))
        cnt = vb["row_loop_common_path"]["counts"]
        cost = vb["class_cost_cycles"]
        issue = cnt["fast"] + cnt["slow"] + cnt["qrate"] + cnt["nop"]
        cyc = sum(cnt[k] * cost[k] for k in cnt)
        valu = {"unit": "G wave-instructions/s", "peak_vector_2cycle": VECTOR_PEAK_GWINST,
                "mix_ceiling": SIMDS * CLOCK_GHZ / (cyc / issue),
                "kernel": vb["kernel"], "class_counts_per_wave_step": cnt, "class_cost_cycles": cost,
                "budget_source": "profiles/valu_budget.json (tools/valu_budget.py over the compiler's assembly)",
                "note": "mix ceiling = issue rate if every instruction of the row loop cost exactly its class cost "
                        "(fast 2, slow 4, s_nop 4 cycles per wave-instruction per SIMD, measured by tools/ubench/valu_rates*.hip)"}
        if vb.get("source_digest") != digest:
            valu["budget_source"] += " -- STALE (digest %s, sources now %s)" % (vb.get("source_digest"), digest)
        if tj and tj.get("source_digest") == digest and "sq_insts_valu_per_launch" in tj and same_shape:
            ginst = tj["sq_insts_valu_per_launch"] * scale / (avg_launch_ms * 1e-3) / 1e9
            valu.update({"achieved": ginst, "insts_per_launch": int(tj["sq_insts_valu_per_launch"] * scale),
                         "frac_of_vector_peak": ginst / VECTOR_PEAK_GWINST, "frac_of_mix_ceiling": ginst / valu["mix_ceiling"],
                         "insts_source": "SQ_INSTS_VALU of " + tj.get("source", "profiles/traffic.json") +
                                         ("" if scale == 1 else " -- EXTRAPOLATED x%.3g" % scale)})
        res["roofline"]["valu"] = valu
    except (OSError, KeyError, ValueError, ZeroDivisionError):
        pass

    # ... and the same counters measured IN THIS RUN (child processes under rocprofv3 --pmc, N = 1): when they succeed they
    # replace the committed figures above, which stay in the line as "committed_profile" for comparison
    if world == 1 and not args.no_live_counters and res["config"]["kernel_kind"] == 1:
        lc = _live_counters(F)
        if "error" in lc or not {"FETCH_SIZE", "WRITE_SIZE"} <= set(lc):
            res["roofline"]["live_counters"] = lc
        else:
            committed = {"traffic": res["roofline"].get("traffic"), "traffic_source": res["roofline"].get("traffic_source")}
            live_bytes = int(lc["FETCH_SIZE"] * 1024 * 2 + lc["WRITE_SIZE"] * 1024)
            res["roofline"]["traffic"] = live_bytes
            res["roofline"]["traffic_source"] = ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate child passes over "
                                                 "bench.py --headline-only, %d pairs per launch, mean of %d launches), FETCH_SIZE x2 (gfx950) + WRITE_SIZE, KB"
                                                 % (F, lc.get("launches_sampled", 0)))
            res["roofline"]["traffic_over_algorithmic"] = live_bytes / alg_bytes_per_launch
            res["roofline"]["committed_profile"] = committed
            res["roofline"]["live_counters"] = lc
            if "SQ_INSTS_VALU" in lc and "valu" in res["roofline"]:
                v = res["roofline"]["valu"]
                ginst = lc["SQ_INSTS_VALU"] / (avg_launch_ms * 1e-3) / 1e9
                v.update({"achieved": ginst, "insts_per_launch": int(lc["SQ_INSTS_VALU"]), "frac_of_vector_peak": ginst / VECTOR_PEAK_GWINST,
                          "frac_of_mix_ceiling": ginst / v["mix_ceiling"], "insts_source": "SQ_INSTS_VALU measured in this run (child pass under rocprofv3 --pmc)",
                          "valu_per_64_candidates": lc["SQ_INSTS_VALU"] / (cands * F / 64.0),
                          "simd_cycles_per_valu_inst": SIMDS * CLOCK_GHZ * 1e9 * (avg_launch_ms * 1e-3) / lc["SQ_INSTS_VALU"]})

    if world == 1 and args.host_frames > 0:
        try:
            res.update(_host_legs(sha, ctx, local, Ls, Rs, out, block, maxdisp, args.flags, min(args.host_frames, F)))
            # the host-fed legs compare their maps with the device-resident launch's: a difference fails the run like any other
            hf = res.get("host_fed", {})
            wrong = [n for n, v in (("host_fed", hf.get("matches_device_resident")), ("host_fed.pageable", hf.get("pageable", {}).get("matches_device_resident")),
                                    ("host_fed.rgba host gray vs device gray", hf.get("single_call_ms", {}).get("rgba_maps_identical")))
                     if v is False] + [f"pool[{i}]" for i, p in enumerate(res.get("pool", [])) if p.get("matches_device_resident") is False]
            if wrong:
                res["parity_error"] = "host-fed result differs from the device-resident one: " + ", ".join(wrong)
        except Exception as e:                                    # side measurements never fail the headline
            res["host_fed"] = {"error": repr(e)}

    if not args.no_cpu_baseline and world == 1:                  # rank 0 at N = 1 only (the contract)
        # the reference's own C (hardware/sad.c, compiled unmodified into oracle/_ref/ in the build
        # container) on its own fixed workload, next to the GPU running the same semantics: not the
        # metric's workload, so reported beside cpu_baseline, not as it
        ref_c = _time_reference_c(ctx, sha)
        if ref_c:
            res["reference_c"] = ref_c
        import oracle
        # CPU baseline: the literal C restatement of pkg/despair sad.go:55-94 on this box's
        # host cores, same frames as the GPU: one thread, then every core this process may use (row bands
        # like output.go:172).  Bounded sample: ~15-25 s of wall time in all.
        # "all cores" = this process's share of the host: affinity / cgroup quota, and at most 16 -- a
        # one-GPU box of this pool shares its host with other tenants and 16 cores is the per-GPU share
        # (256 threads on the shared 128-core host measured SLOWER than 16: 3.1 vs 3.7 pairs/s)
        ncpu = args.cpu_threads or min(_usable_cpus(), 16)
        host_cpus = os.cpu_count() or ncpu
        model = _cpu_model()

        def cpu_run(npairs, nthr):
            wall, ok = 0.0, True
            for i in range(npairs):
                L0, R0 = _to_host(Ls[i]), _to_host(Rs[i])
                t1 = time.perf_counter()
                cpu = oracle.disparity_rows_mt(L0, R0, block, maxdisp, 0, H, nthr)  # one full pair
                wall += time.perf_counter() - t1
                ok = ok and bool(np.array_equal(cpu, _to_host(out[i])))
            return wall, ok

        n1 = 1 if F >= 1 else 0                                   # one full pair single-threaded (~5-12 s)
        w1, ok1 = cpu_run(n1, 1)
        nn = min(F, max(2, min(64, ncpu)))                       # ~4-5 s of wall time at nproc threads (one pair per thread-equivalent)
        wn, okn = cpu_run(nn, ncpu)
        ok = ok1 and okn
        res["cpu_baseline"] = {
            "value": nn / wn, "unit": "pairs/s", "cores": ncpu, "kind": "port", "cpu_model": model, "host_logical_cpus": host_cpus,
            # "port" because the reference itself (Go) cannot run here: probed in this run, not assumed
            "go_toolchain": _go_toolchain(),
            "sample": f"{nn} full {W}x{H} pairs (frame seeds 1000..{1000 + nn - 1} of this rank), literal restatement of "
                      f"pkg/despair sad.go:55-94 in C, {ncpu} threads over output.go row bands, {wn:.2f} s wall = {wn * ncpu:.0f} CPU-seconds"
                      + ("" if args.cpu_threads else f"; {ncpu} = this tenant's share of the {host_cpus}-thread host -- the all-thread run was measured in round 2 "
                                                     "(256 threads on the shared host) and is SLOWER: 3.1 against 3.7 pairs/s"),
            "single_thread": {"value": n1 / w1 if w1 > 0 else None, "unit": "pairs/s", "cores": 1,
                              "sample": f"{n1} full pair, same code, 1 thread, {w1:.2f} s"},
            "gpu_output_matches": ok}
        if not ok:
            res["parity_error"] = "a GPU frame differs from the CPU oracle"
    code = _emit(res)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    if code:
        sys.exit(code)


if __name__ == "__main__":
    main()
