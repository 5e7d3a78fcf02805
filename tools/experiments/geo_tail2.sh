cd $GRAFT_REPO_ROOT
cat > /tmp/geo.py <<'PY'
import sys, os
sys.path.insert(0, ".")
import torch
import steroscopic_hardware_amd as sha
sha.capi.LIB_PATH = os.path.join(os.path.dirname(sha.capi.LIB_PATH), "libsadhip_tune.so")
torch.zeros(1, device="cuda")
for c in ((3840,2160,15,128,4),(3840,2160,15,128,8),(3840,2160,15,128,16),(3840,2160,15,128,32),(1920,1080,15,128,64),(1920,1080,13,128,32),(1920,1080,16,64,64)):
    g = sha.capi.plan_describe_batch(*c)
    print(os.environ.get("SADHIP_TUNE_TAIL2"), c, [(p["strips"], p["rows"], p["wg_per_cu"], p["tiles"]) for p in g], flush=True)
PY
SADHIP_TUNE_TAIL2=0 python /tmp/geo.py; SADHIP_TUNE_TAIL2=1 python /tmp/geo.py
for s in 8 9 10 11 12 13; do echo -n "c5/16 strips=$s "; SADHIP_TUNE_STRIPS=$s python tools/bench_cfg.py 3840 2160 15 128 16 2>/dev/null | grep us/frame; done
for s in 3 4 5 6; do echo -n "B15 D128 64 strips=$s "; SADHIP_TUNE_STRIPS=$s python tools/bench_cfg.py 1920 1080 15 128 64 2>/dev/null | grep us/frame; done
