"""DESIGN.md §4's per-launch table from the newest recording under profiles/ (kernel stats, HBM traffic, SQ counters):
python scripts/design_table.py [round, default 05] [version, default 1]"""
import csv, os, sys
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "06"
v = sys.argv[2] if len(sys.argv) > 2 else "1"


def col(path, key, sub, field):
    for r in csv.DictReader(open(os.path.join(here, "profiles", path))):
        if sub in r[key]:
            return r, float(r[field]) if field else None
    raise KeyError(sub)


ROWS = [  # label, kernel substring, flops (G), piece products, bytes (MB)
    ("first layer forward (`obs_fwd_h2_kernel`)", "obs_fwd_h2", 107.4, 2, 1301),
    ("first layer weight gradient (`obs_bwd_h2_kernel`)", "obs_bwd_h2", 107.4, 2, 1301),
    ("conv2 forward (`h2conv_kernel<0,3>`)", "h2conv_kernel<0, 3>", 87.0, 3, 1179),
    ("conv2 data gradient (`h2conv_kernel<3,2>`)", "h2conv_kernel<3, 2>", 87.0, 3, 1205),
    ("conv2 weight gradient (`h2wgrad_kernel<0,2>`)", "h2wgrad_kernel<0, 2>", 87.0, 3, 1179),
    ("conv3 forward (`h2conv_kernel<1,3>`)", "h2conv_kernel<1, 3>", 59.2, 3, 545),
    ("conv3 data gradient (`h2conv_kernel<2,2>`)", "h2conv_kernel<2, 2>", 59.2, 3, 545),
    ("conv3 weight gradient (`h2wgrad_kernel<1,3>`)", "h2wgrad_kernel<1, 3>", 59.2, 3, 545),
    ("Linear forward (`h2gemm_kernel<4,0,3>`)", "h2gemm_kernel<4, 0, 3, true", 52.6, 3, 245),
    ("Linear data gradient (`h2gemm_kernel<8,0,2,false>`)", "h2gemm_kernel<8, 0, 2, false", 52.6, 3, 245),
    ("Linear weight gradient (`h2tn_kernel`, 4 row ranges + slab sum)" if rnd >= "06" else
     "Linear weight gradient (`gemm3_kernel<128,128,…,2>`, split-K 5)", "h2tn_kernel" if rnd >= "06" else "gemm3_kernel<128, 128", 52.6, 3, 245),
]
print("| launch (kernel) | flops | MFMA floor | bytes | HBM floor | measured | × larger floor | traffic | MFMA busy | vector busy |")
print("|---|---|---|---|---|---|---|---|---|---|")
total = 0.0
for label, sub, gf, pp, mb in ROWS:
    k, ns = col(f"r{rnd}_bench_kernel_stats_v{v}.csv", "Name", sub, "AverageNs")
    t, _ = col(f"r{rnd}_hbm_traffic_v{v}.csv", "kernel", sub, None)
    c, _ = col(f"r{rnd}_sq_counters_v{v}.csv", "kernel", sub, None)
    us = ns / 1e3
    mf, hf = gf * pp / 2.5, mb / 8.0   # us: G flops x products / 2.5 PFLOP/s; MB / 8 TB/s
    traffic = (float(t["FETCH_bytes_per_launch_corrected_x2"]) + float(t["WRITE_bytes_per_launch"])) / 1e6
    total += us
    extra = ""
    print(f"| {label} | {gf} G | {mf:.0f} µs{' (×2)' if pp == 2 else ''} | {mb} MB | {hf:.0f} µs | {us:.0f} µs{extra} | {us / max(mf, hf):.1f} | "
          f"{traffic:.0f} MB | {float(c['mfma_busy']):.2f} | {float(c['valu_busy']):.2f} |")
print(f"\nsum of the eleven launches: {total / 1e3:.2f} ms per chunk = {total * 32 / 1e3:.1f} ms per update")
alg = sum(r[4] for r in ROWS)
tr = 0.0
for label, sub, gf, pp, mb in ROWS:
    t, _ = col(f"r{rnd}_hbm_traffic_v{v}.csv", "kernel", sub, None)
    tr += (float(t["FETCH_bytes_per_launch_corrected_x2"]) + float(t["WRITE_bytes_per_launch"])) / 1e6
print(f"algorithmic bytes of the eleven launches: {alg / 1e3:.2f} GB per chunk = {alg * 32 / 1e3:.0f} GB per update; counted traffic {tr / 1e3:.2f} GB per chunk = "
      f"{tr * 32 / 1e3:.0f} GB per update = {tr / alg:.2f} x")
