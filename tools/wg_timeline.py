#!/usr/bin/env python3
"""Round 5 diagnostic: WHICH CU ran WHICH workgroup WHEN.  Runs one configuration's batched launch on the stamps build
(lib/libsadhip_stamps.so, make -C csrc stamps) with SADHIP_TUNE_WGTL_FILE set -- every workgroup records its start and end
(s_memrealtime: one 100 MHz clock for the whole chip) and its HW_ID / XCC_ID -- and accounts for the launch's slot-time:
busy, idle before a CU's first workgroup, idle in gaps between workgroups, idle after a CU's last workgroup (the tail), and
how much workgroup durations differ by tile.  usage (GPU box): tools/wg_timeline.py W H BLOCK MAXDISP NFRAMES [strips]"""
import csv, os, sys, collections, statistics
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
W, H, B, D, N = (int(v) for v in sys.argv[1:6])
out = os.path.join(REPO, "gpurun_out", f"wgtl_{W}x{H}_b{B}_d{D}_n{N}" + (f"_s{sys.argv[6]}" if len(sys.argv) > 6 else "") + ".csv")
if os.path.exists(out): os.remove(out)
os.environ["SADHIP_TUNE_WGTL_FILE"] = out
if len(sys.argv) > 6: os.environ["SADHIP_TUNE_STRIPS"] = sys.argv[6]
import torch
import steroscopic_hardware_amd as sha
sha.capi.LIB_PATH = os.path.join(os.path.dirname(sha.capi.LIB_PATH), "libsadhip_stamps.so")
parts = [sha.synth.synth_batch_torch([1000 + i for i in range(c, min(c + 32, N))], W, H, "cuda:0") for c in range(0, N, 32)]
Ls, Rs = torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts]); del parts
o = torch.empty_like(Ls)
ctx = sha.capi.Context(0, W, H, 1)
for _ in range(3): ctx.disparity_dev(Ls, Rs, o, B, D); torch.cuda.synchronize()
rows = list(csv.DictReader(open(out)))
last = max(int(r["launch"]) for r in rows)
rows = [r for r in rows if int(r["launch"]) == last and int(r["t1"]) > 0]
t_begin = min(int(r["t0"]) for r in rows); t_end = max(int(r["t1"]) for r in rows)
span = t_end - t_begin
cus = collections.defaultdict(list)
for r in rows:
    hw, xcc = int(r["hw_id"]), int(r["xcc_id"]) & 0xf
    cu = (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf)
    cus[cu].append((int(r["t0"]) - t_begin, int(r["t1"]) - t_begin, int(r["y"])))
r0 = rows[0]
print(f"{W}x{H} b{B} D{D} n={N}: kernel P{r0['p']} ND{r0['nd']} H{r0['h']} nw{r0['nw']}, {r0['strips']} strips of {r0['rows']} rows, {len(rows)} workgroups on {len(cus)} CUs, "
      f"launch span {span / 100:.1f} us ({span / 100 / N:.1f} us per pair)")
per_cu = max(max(sum(1 for a in v if a[0] <= t < a[1]) for t in [x[0] for x in v]) for v in cus.values())
busy = head = gaps = tail = 0
for v in cus.values():
    v.sort()
    slots = [[] for _ in range(per_cu)]
    for a in v:                                                # greedy: the slot that has been free longest
        k = min(range(per_cu), key=lambda i: slots[i][-1][1] if slots[i] else -1)
        slots[k].append(a)
    for sl in slots:
        if not sl: head += span; continue
        head += sl[0][0]; tail += span - sl[-1][1]
        busy += sum(b - a for a, b, _ in sl)
        gaps += sum(max(0, sl[i + 1][0] - sl[i][1]) for i in range(len(sl) - 1))
tot = span * len(cus) * per_cu
print(f"  resident workgroups per CU at most {per_cu}; slot-time: busy {100 * busy / tot:.1f} %, idle before the first workgroup {100 * head / tot:.1f} %, "
      f"in gaps between workgroups {100 * gaps / tot:.1f} %, after the last one (tail) {100 * tail / tot:.1f} %")
dur = collections.defaultdict(list)
for v in cus.values():
    for a, b, y in v: dur[y].append((b - a) / 100)
alld = [d for v in dur.values() for d in v]
print(f"  workgroup duration: mean {statistics.mean(alld):.1f} us, min {min(alld):.1f}, max {max(alld):.1f}; by tile (blockIdx.y): " +
      " ".join(f"{y}:{statistics.mean(v):.0f}" for y, v in sorted(dur.items())))
ends = sorted(b for v in cus.values() for _, b, _ in v)
n = len(ends)
print(f"  last workgroup ends at {span / 100:.1f} us; 90 % of the workgroups have ended by {ends[int(0.9 * n)] / 100:.1f} us, 99 % by {ends[int(0.99 * n)] / 100:.1f} us")
starts = sorted(a for v in cus.values() for a, _, _ in v)
print(f"  first round: {sum(1 for s in starts if s < 500)} workgroups started within the first 5 us, {sum(1 for s in starts if s < 2000)} within 20 us")
ctx.close()
