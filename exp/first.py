import sys, pandas as pd
for name in sys.argv[1:]:
    d = pd.read_csv(f"gpurun_out/exp_{name}/t_kernel_trace.csv")
    d["dur"] = (d["End_Timestamp"] - d["Start_Timestamp"]) / 1e3
    d["k"] = d["Kernel_Name"].str.replace(r"\(.*", "", regex=True).str.replace("osh::", "").str.replace("void ", "")
    first = d.groupby("k").first()["dur"]
    mx = d.groupby("k")["dur"].max()
    print(name, {k: round(float(mx[k]), 1) for k in mx.index if "rocclr" not in k and mx[k] > 50})
