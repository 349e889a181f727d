"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV (one steady-state step):
   python tools/trace_gaps.py <kernel_trace.csv>"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
conv = [i for i, r in enumerate(rows) if "k_conv" in r["Kernel_Name"]]
if len(conv) < 3:
    sys.exit("need at least three steps in the trace")
a, b = conv[-2], conv[-1]          # one full step: from the start of a convolution to the start of the next
span = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[a:b]) / 1e3
gaps = [(int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"])) / 1e3 for i in range(a, b)]
big = sorted(((g, rows[a + i]["Kernel_Name"][:40], rows[a + i + 1]["Kernel_Name"][:40]) for i, g in enumerate(gaps)), reverse=True)[:6]
print(f"step {span:.1f} us, kernels {busy:.1f} us, idle {span - busy:.1f} us over {len(gaps)} boundaries (median gap {sorted(gaps)[len(gaps) // 2]:.2f} us)")
for g, x, y in big:
    print(f"   {g:9.2f} us  between {x} -> {y}")
