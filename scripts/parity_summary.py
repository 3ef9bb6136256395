"""Summarise gpurun_out/parity_report.jsonl (written by the -m gpu tests) into profiles/<round>/parity_errors.json (`parity_summary.py r03`; default r02): per kind
of test the number of cases, the worst max-normalised and element-wise errors, and every case that needed more than the 1e-5
of BASELINE.json's north_star (with the reference's own fp32-vs-fp64 gap on that bank beside it)."""
import collections, json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = [json.loads(l) for l in open(os.path.join(REPO, "gpurun_out", "parity_report.jsonl"))]
out = {"source": "python -m pytest tests -m gpu on an MI355X box; tests/helpers.py report()", "cases": len(rows), "by_kind": {}}
by = collections.defaultdict(list)
for r in rows:
    by[r["kind"]].append(r)
for k, rs in by.items():
    e = [r for r in rs if "err_max_norm" in r]
    d = {"cases": len(rs)}
    if e:
        w = max(e, key=lambda r: r["err_max_norm"])
        we = max(e, key=lambda r: r.get("err_elementwise", 0))
        d.update(worst_max_normalised=w["err_max_norm"], worst_case={x: w.get(x) for x in ("B", "G", "Cq", "H", "W", "K", "shape", "variant") if x in w},
                 worst_elementwise=we.get("err_elementwise"),
                 elementwise_note="max |a-b| / max(|b|, 1e-3 max|b|): elements down to a thousandth of the largest are judged against themselves",
                 over_1e5=[{x: r.get(x) for x in ("B", "G", "Cq", "H", "W", "K", "err_max_norm", "reference_fp32_vs_fp64", "tol")}
                                        for r in e if r.get("needed_more_than_1e5")])
        forms = collections.Counter(str((r.get("variant") or {}).get("sec")) for r in e if r.get("variant"))
        if forms:
            d["io_forms_launched"] = dict(forms)
    else:
        d["records"] = rs
    out["by_kind"][k] = d
path = os.path.join(REPO, "profiles", sys.argv[1] if len(sys.argv) > 1 else "r02", "parity_errors.json")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps({k: {x: v[x] for x in v if x in ("cases", "worst_max_normalised", "worst_elementwise", "io_forms_launched")} for k, v in out["by_kind"].items()}, indent=1))
