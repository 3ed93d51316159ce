"""bench.py's step against the trainer's chunk_rows (rows per forward/backward launch group): wave quantisation of the
tile grids (e.g. the FC forward's 128x128 tiles: 512 per 16384 rows on 768 workgroup slots) against activation memory."""
import json
import subprocess
import sys

for c in [int(a) for a in sys.argv[1:]] or [16384, 24576, 32768, 40960]:
    out = subprocess.run([sys.executable, "bench.py", "--steps", "5", "--warmup", "2", "--chunk-rows", str(c), "--no-cpu-baseline",
                          "--no-from-host"], capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(c, "failed", out.stderr[-400:])
        continue
    d = json.loads(line[-1])
    k = d["kernel_ms_per_step"]
    print(f"chunk_rows {c}: {d['ms_per_step']:.2f} ms/step, {d['value'] / 1e6:.3f} M env-steps/s; " +
          ", ".join(f"{n} {v:.1f}" for n, v in list(k.items())[:7]), flush=True)
