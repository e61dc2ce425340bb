#!/usr/bin/env python3
"""Turn the JSON-lines log of tests/test_gpu_parity.py ($OIVA_PARITY_LOG) into the markdown table committed
under profiles/.   python tools/parity_table.py gpurun_out/parity.jsonl > profiles/r05_parity_errors.md"""
import json
import sys


def f(x):
    return "-" if x is None else f"{x:.1e}"


rows = [json.loads(line) for line in open(sys.argv[1])]
e2e = [r for r in rows if r["test"] == "e2e"]
print("# Achieved parity errors (MI355X, round 6)\n")
print("Source: `tests/test_gpu_parity.py` run with `OIVA_PARITY_LOG` on the GPU box; distances are relative Frobenius")
print("norms.  `floor` = distance between the REAL reference's complex64 and complex128 results on the fixture")
print("(stored by `tests/golden/make_golden.py`); `amp` = the reference's own amplification of a 1e-12 input")
print("perturbation.  Default arithmetic (`auto`): complex128 input runs `precise`; complex64 input runs `precise` too on frame axes")
print("up to 256 long with up to 8 channels (the reference forms those covariances in complex128 as well, overiva.py:179; 2, 6, 8")
print("channels with 1-2 sources whose X fits on chip keep `mixed` and the X-resident kernel), `mixed` elsewhere (float32 products")
print("and lane chains, float64 sums and per-bin algebra).  The `mode` column says which ran.  `fast` = float32 per-bin algebra too.\n")
print("## overiva(), complex64 input (the default mode of that input), final W after n_iter iterations\n")
print("`jitter` = how far the reference's own complex64 W moves when X changes in its last bit (tests/golden/c64_jitter.npz); rows")
print("whose jitter exceeds 1e-3 are held to max(floor, jitter) instead of the floor (marked *).  Rows more than one floor from the")
print("complex128 result would be marked +: round 5 had one (z_iid gauss 20, `mixed`), round 6 none.\n")
print("| fixture | model | n_iter | amp | mode | reference c64 floor | c64 jitter | W vs reference-c64 | in floors | W vs c128 | in floors | Y vs c128 | fast: W vs c128 | in floors |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
key = lambda r: (r["fixture"], r["model"], r["n_iter"])
dflt = {key(r): r for r in e2e if r["input"] == "c64" and r["mode"] in ("mixed", "precise")}
fast = {key(r): r for r in e2e if r["mode"] == "fast"}
worst = {}
for k in sorted(dflt):
    p, q = dflt[k], fast.get(k)
    fl = p.get("ref_c64_floor")
    r64 = p["W_vs_ref_c64"] / fl if fl and p.get("W_vs_ref_c64") is not None else float("nan")
    r128 = p["W_vs_c128"] / fl if fl else float("nan")
    if fl and fl > 2e-7:
        w = worst.setdefault(p["mode"], [0.0, 0.0])
        w[0], w[1] = max(w[0], r64), max(w[1], r128)
    jit = p.get("ref_c64_jitter")
    over = "+" if p["mode"] == "mixed" and fl and fl > 2e-7 and r128 > 1.0 and not (jit and jit > 1e-3) else ""
    print(f"| {k[0]} | {k[1]} | {k[2]} | {p['amp']:.1f} | {p['mode']} | {f(fl)} | {f(jit)}{'*' if jit and jit > 1e-3 else ''} | {f(p.get('W_vs_ref_c64'))} | {r64:.2f} | {f(p['W_vs_c128'])} | {r128:.2f}{over} | "
          f"{f(p.get('Y_vs_c128'))} | {f(q['W_vs_c128']) if q else '-'} | {(q['W_vs_c128'] / fl if q and fl else float('nan')):.1f} |")
print("\nWorst ratios where the floor exceeds 2e-7 (distance to the reference's complex64 result / to its complex128 result, in floors): "
      + "; ".join(f"{m}: {w[0]:.2f} / {w[1]:.2f}" for m, w in sorted(worst.items())))
print("\n## overiva(), complex128 input (`precise`), final W\n")
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
