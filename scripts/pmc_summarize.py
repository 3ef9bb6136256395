"""Summarise rocprofv3 --pmc passes: per kernel, mean counter value per dispatch."""
import csv, glob, os, sys, collections, json
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "finc_wave_kernel" in k:
            k = "inverse"
        elif "finc_split_kernel" in k:
            k = "inverse_split"
        elif "finc_chain_kernel" in k:
            k = "inverse_chain"
        elif "finc_conv_kernel" in k:
            k = "forward"
        elif "finc_wino4m_kernel" in k:
            k = "forward_wino4_msplit"
        elif "finc_wino4_kernel" in k:
            k = "forward_wino4"
        elif "finc_wino_kernel" in k:
            k = "forward_wino"
        elif "finc_wino5_kernel" in k:
            k = "forward_wino5"
        elif "finc_gradw_wino_kernel" in k:
            k = "gradw_wino"
        elif "finc_gradw_winot_kernel" in k:
            k = "gradw_wino_tiled"
        elif "finc_gradw_staged_kernel" in k:
            k = "gradw_staged"
        elif "finc_gradw_tiled_kernel" in k:
            k = "gradw_tiled"
        elif "finc_stream_kernel" in k:
            k = "stream_inverse" if "ELb1ELb" in k or "true, " in k.split("finc_stream_kernel")[1][:24] else "stream_forward"
        elif "finc_big_kernel" in k:
            k = "inverse_big"
        elif "finc_bigfwd_kernel" in k:
            k = "forward_big"
        else:
            continue
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
out["n_dispatches"] = {k: {c: len(v) for c, v in d.items()} for k, d in acc.items()}
json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
