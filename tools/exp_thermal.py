"""Experiment: the 5x5 Gaussian's launch time over a minute of back-to-back launches, beside the card's sensors
(hwmon temperatures, clocks, power) — is the 1.10 vs 1.25 ms spread a thermal / power state?
usage: python tools/exp_thermal.py [seconds] [frames]"""
import glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from imagetransformations_amd import _ffi
SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 45
F = int(sys.argv[2]) if len(sys.argv) > 2 else 128
H, W = 2160, 3840
dev = torch.device("cuda:0")
cur = torch.cuda.current_stream()
s = torch.randint(0, 256, (F, H, W, 3), dtype=torch.uint8, device=dev)
d = torch.empty_like(s)
vs, vd = _ffi.view_of(s), _ffi.view_of(d)
def go(): _ffi.call("imgxf_gaussian_u8", _ffi.vp(vs), _ffi.vp(vd), 5, 5.0 / 6.0, None, cur.cuda_stream)
def sensors():
    out = {}
    for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for f in sorted(glob.glob(hw + "/temp*_input") + glob.glob(hw + "/freq*_input") + glob.glob(hw + "/power*_average") + glob.glob(hw + "/power*_input")):
            try:
                lab = f.replace("_input", "_label").replace("_average", "_label")
                name = open(lab).read().strip() if os.path.exists(lab) else os.path.basename(f)
                out[os.path.basename(os.path.dirname(hw))[:0] + name] = int(open(f).read().strip())
            except Exception:
                pass
        break
    return out
print("sensors:", sensors(), flush=True)
t0 = time.time()
k = 0
while time.time() - t0 < SECS:
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(cur)
    for _ in range(200): go()
    b.record(cur); torch.cuda.synchronize()
    sn = sensors()
    print(f"t={time.time() - t0:5.1f}s  {a.elapsed_time(b) / 200:.4f} ms/launch  " + "  ".join(f"{k_}={v}" for k_, v in sn.items()), flush=True)
    k += 1
    if k == 20:                       # a pause: does the card recover when idle?
        print("idle 8 s", flush=True); time.sleep(8)
