import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "oracle")]
import numpy as np
import steroscopic_hardware_amd as sha, oracle
if os.environ.get("SADHIP_LIB"): sha.capi.LIB_PATH = os.path.abspath(os.environ["SADHIP_LIB"])
for (w, h, B, D) in ((400, 150, 16, 64), (1920, 40, 16, 64), (640, 60, 17, 32)):
    L, R = sha.synth.synth_pair(43, w, h)
    with sha.capi.Context(0, w, h, 1) as c:
        got = c.disparity(L, R, B, D, sha.capi.FLAG_RAW)
        exp = c.disparity(L, R, B, D, sha.capi.FLAG_RAW | sha.capi.FLAG_FORCE_GENERIC)
    bad = np.argwhere(got != exp)
    print(os.environ.get("SADHIP_LIB", "product"), (w, h, B, D), "mismatches", len(bad), "rows", sorted(set(bad[:, 0]))[:12], "cols", (bad[:, 1].min(), bad[:, 1].max()) if len(bad) else None)
    if len(bad): print("   first", [(int(y), int(x), int(got[y, x]), int(exp[y, x])) for y, x in bad[:6]])
