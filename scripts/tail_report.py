"""Who is late at the tail of an update -- the host or the GPU?  From a rocprofv3 --kernel-trace --hip-trace run of the benchmark:
for the last launches of the last update (up to the optimiser kernel), the time between the host's launch call and the kernel's
start on the GPU, and the host's HIP calls of that window that took longer than 20 us.
    rocprofv3 --kernel-trace --hip-trace --output-format csv -d /tmp/kt -- python3 bench.py ...; python3 scripts/tail_report.py /tmp/kt [n]"""
import csv, glob, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
ht = glob.glob(d + "/**/*hip_api_trace.csv", recursive=True)[0]
K = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Correlation_Id"]) for r in csv.DictReader(open(kt))))
H = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], r["Correlation_Id"]) for r in csv.DictReader(open(ht))]
by_corr = {h[3]: h for h in H}
ends = [i for i, k in enumerate(K) if "adam" in k[2].lower()]
seg = K[ends[-2] + 1:ends[-1] + 1][-n:]
t0 = seg[0][0]
print("kernel start (us, relative)   duration   launch call -> start   name")
for s, e, name, corr in seg:
    h = by_corr.get(corr)
    lag = f"{(s - h[1]) / 1e3:9.1f}" if h else "        ?"
    print(f"  +{(s - t0) / 1e3:9.1f}  {(e - s) / 1e3:8.1f}  {lag}   {name[:80]}")
w0, w1 = seg[0][0] - 3_000_000, seg[-1][1]
print("host HIP calls longer than 20 us in the 3 ms before / during that window:")
for s, e, fn, corr in sorted(H):
    if w0 <= s <= w1 and e - s > 20_000:
        print(f"  +{(s - t0) / 1e3:9.1f}  {(e - s) / 1e3:8.1f} us  {fn}")
