"""Engine clock and socket power of each encoder chain alone (DEV library, HCM_SKIP mask: 0 = whole step, 12 = RGB trunks only,
7 = BERT only, 11 = depth trunks only): ~6 s of act() steps per mask with rocm-smi sampled beside them.
usage (GPU box): python tools/chain_power.py [B]   (needs `make DEV=1`)"""
import os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = sys.argv[1] if len(sys.argv) > 1 else "64"
CHILD = r'''
import os, sys, time, torch
sys.path.insert(0, %r)
import hcm_pkg; hcm_pkg.load()
from robo_vln_amd import synth
from robo_vln_amd.config import HCMConfig
from robo_vln_amd.policy import HCMEngine
B = int(sys.argv[1])
cfg = HCMConfig().validate()
hi, lo = synth.make_weights(cfg, seed=0)
eng = HCMEngine(cfg, hi, lo, max_batch=B, precision="fp16", graph=True)
obs = {k: torch.from_numpy(v).cuda() for k, v in synth.make_observations(cfg, B, rgb_uint8=True).items()}
R = cfg.num_recurrent_layers
hh = torch.zeros(R, B, cfg.hidden, device="cuda"); lh = torch.zeros(R, B, cfg.hidden, device="cuda"); m = torch.ones(B, device="cuda")
for _ in range(6): eng.act(obs, hh, lh, m)
torch.cuda.synchronize()
print("READY", flush=True)
t0 = time.time(); n = 0
while time.time() - t0 < 7.0:
    for _ in range(50): eng.act(obs, hh, lh, m)
    torch.cuda.synchronize(); n += 50
print("MS_PER_STEP %%.3f" %% ((time.time() - t0) / n * 1e3), flush=True)
''' % ROOT


def sample():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    clk = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
    pw = re.search(r"Power \(W\): ([\d.]+)", out)
    return (int(clk.group(1)) if clk else -1, float(pw.group(1)) if pw else -1.0)


for mask, name in ((0, "whole step"), (12, "RGB pair trunk only"), (7, "BERT only"), (11, "depth pair trunk only")):
    env = dict(os.environ, HCM_DEV_LIB="1", HCM_SKIP=str(mask))
    p = subprocess.Popen([sys.executable, "-c", CHILD, B], env=env, stdout=subprocess.PIPE, text=True)
    assert p.stdout.readline().strip() == "READY"
    time.sleep(1.5)
    s = []
    while p.poll() is None and len(s) < 14:
        s.append(sample()); time.sleep(0.25)
    ms = p.stdout.read().strip()
    p.wait()
    s = [x for x in s if x[1] > 0]
    print(f"| {name} | {ms.replace('MS_PER_STEP ', '')} | {min(x[0] for x in s)}-{max(x[0] for x in s)} | {sum(x[1] for x in s) / len(s):.0f} |", flush=True)
