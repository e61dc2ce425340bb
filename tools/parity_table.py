#!/usr/bin/env python3
"""Turn the JSON-lines log of tests/test_gpu_parity.py ($OIVA_PARITY_LOG) into the markdown table committed
under profiles/.   python tools/parity_table.py gpurun_out/r2b/parity.jsonl > profiles/r02_parity_errors.md"""
import json
import sys


def f(x):
    return "-" if x is None else f"{x:.1e}"


rows = [json.loads(line) for line in open(sys.argv[1])]
e2e = [r for r in rows if r["test"] == "e2e"]
print("# Achieved parity errors (MI355X, round 2)\n")
print("Source: `tests/test_gpu_parity.py` run with `OIVA_PARITY_LOG` on the GPU box; distances are relative Frobenius")
print("norms.  `floor` = distance between the REAL reference's complex64 and complex128 results on the fixture")
print("(stored by `tests/golden/make_golden.py`); `amp` = the reference's own amplification of a 1e-12 input")
print("perturbation.  precise = float64 covariance + float64 per-bin algebra (default); fast = float32 everywhere.\n")
print("## overiva(), complex64 input, final W after n_iter iterations\n")
print("| fixture | model | n_iter | amp | reference c64 floor | precise: W vs c128 | precise: W vs reference-c64 | precise: Y vs c128 | fast: W vs c128 | fast, in floors |")
print("|---|---|---|---|---|---|---|---|---|---|")
key = lambda r: (r["fixture"], r["model"], r["n_iter"])
prec = {key(r): r for r in e2e if r["mode"] == "precise" and r["input"] == "c64"}
fast = {key(r): r for r in e2e if r["mode"] == "fast"}
for k in sorted(prec):
    p, q = prec[k], fast.get(k)
    fl = p.get("ref_c64_floor")
    print(f"| {k[0]} | {k[1]} | {k[2]} | {p['amp']:.1f} | {f(fl)} | {f(p['W_vs_c128'])} | {f(p.get('W_vs_ref_c64'))} | {f(p.get('Y_vs_c128'))} | "
          f"{f(q['W_vs_c128']) if q else '-'} | {(q['W_vs_c128'] / fl if q and fl else float('nan')):.1f} |")
print("\n## overiva(), complex128 input (precise), final W\n")
print("| fixture | model | n_iter | amp | W vs c128 | Y vs c128 | bound |")
print("|---|---|---|---|---|---|---|")
for r in sorted((r for r in e2e if r["input"] == "c128"), key=key):
    print(f"| {r['fixture']} | {r['model']} | {r['n_iter']} | {r['amp']:.1f} | {f(r['W_vs_c128'])} | {f(r['Y_vs_c128'])} | {f(r['bound_c128'])} |")
print("\n## Other end-to-end checks\n")
print("| test | case | model | n_iter | mode | errors |")
print("|---|---|---|---|---|---|")
for r in rows:
    if r["test"] in ("e2e", "ip_update"):
        continue
    errs = ", ".join(f"{k} {f(v)}" for k, v in r.items() if isinstance(v, float) and k not in ("amp", "bound_c128"))
    print(f"| {r['test']} | {r.get('fixture', '')} | {r.get('model', '')} | {r.get('n_iter', '')} | {r.get('mode', '')} | {errs} |")
ip = [r for r in rows if r["test"] == "ip_update"]
if ip:
    print("\n## Per-bin update kernel from the reference's traced state (W_hat after one epoch)\n")
    print("| arithmetic | lane layout | worst error over fixtures/models/epochs |")
    print("|---|---|---|")
    for fp64 in (False, True):
        for rws in (False, True):
            v = [r["What_err"] for r in ip if r["fp64"] == fp64 and r["rows"] == rws]
            if v:
                print(f"| {'float64' if fp64 else 'float32'} | {'row per lane' if rws else 'element per lane'} | {max(v):.1e} |")
