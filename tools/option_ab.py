"""A/B of a library option under bench.py conditions: usage tools/option_ab.py <option> <v0> <v1> [repeats]."""
import json, os, subprocess, sys
opt, v0, v1 = sys.argv[1], sys.argv[2], sys.argv[3]
rep = int(sys.argv[4]) if len(sys.argv) > 4 else 3
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for r in range(rep):
    for v in (v0, v1):
        env = dict(os.environ, CED_OPTIONS=f"{opt}={v}")
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "40", "--warmup", "5", "--no-cpu-baseline"],
                             env=env, capture_output=True, text=True, timeout=300).stdout
        j = [json.loads(l) for l in out.splitlines() if l.startswith("{")][-1]
        print(f"{opt}={v}: {j['value']/1e9:.4f} Gsamples/s  {j['ms_per_step']:.3f} ms/step", flush=True)
