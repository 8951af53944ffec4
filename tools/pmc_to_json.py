"""profiles/rNN_pmc_dispatches.csv (tools/pmc_summary.py, per-dispatch form) + the ABI-call log of the same forward
(tools/trace_forward.py with BDM_ABI_LOG) -> profiles/rNN_pmc_traffic.json keyed like bench.py's roofline rows ("bdm_function(shape)").

    pmc_to_json.py <dispatches.csv> <abi_log.json> <commit> <out.json>

Every entry is ONE (function, argument shape): the dispatches of a kernel are attributed to the ABI calls that launch that kernel
BY ORDER (the i-th dispatch of `sparse_gemm_s3_kernel` in the marked forward belongs to the i-th bdm_sparse_conv_gemm_s3 call of the
log), so two shapes that share a (kernel, grid) never share an entry.  A function whose calls and dispatches do
not pair up one to one is reported under "unmatched" instead of being averaged."""
import collections, csv, json, sys

# C-ABI function -> (prefix of the rocprof kernel name it launches ONCE per call, algorithmic HBM bytes per launch from the int args | None)
KERNEL_OF = {
    "bdm_conv3d_3x3x3_h2_gn": ("conv3d_h2q_kernel", lambda a: 4 * a[0] * a[3] ** 3 * (a[1] + a[2])),
    "bdm_conv3d_3x3x3_h2": ("conv3d_h2q_kernel", lambda a: 4 * a[0] * a[3] ** 3 * (a[1] + a[2])),
    "bdm_sparse_conv_os_gn": ("sconv_os_kernel", lambda a: 4 * a[0] * a[3] ** 3 * a[2]),       # output grid written once (+ occupied rows, small)
    "bdm_sparse_conv_os": ("sconv_os_kernel", lambda a: 4 * a[0] * a[3] ** 3 * a[2]),
    "bdm_sparse_conv_dil_gn": ("sconv_dil_kernel", lambda a: 4 * a[0] * a[3] ** 3 * a[2]),
    "bdm_sparse_conv_dil": ("sconv_dil_kernel", lambda a: 4 * a[0] * a[3] ** 3 * a[2]),
    "bdm_sparse_conv_gemm_s3": ("sparse_gemm_s3_kernel", None),
    "bdm_sparse_conv_gemm_h2": ("sparse_gemm_h2_kernel", None),
    "bdm_sparse_conv_gather_gn": ("sparse_gather_v4_kernel", lambda a: 4 * a[0] * a[1] * a[2] ** 3),
    "bdm_sparse_conv_gather": ("sparse_gather_v4_kernel", lambda a: 4 * a[0] * a[1] * a[2] ** 3),
    "bdm_sparse_conv_rows_from_map": ("sparse_rows_from_map_kernel", None),
    "bdm_attention_core_h2": ("attn_flash_h2_kernel", None),
    "bdm_ball_query": ("ball_query_kernel", None),
}
disp = list(csv.DictReader(open(sys.argv[1])))
log = json.load(open(sys.argv[2]))   # [[function, [int args...]], ...] in call order
out = {"commit": sys.argv[3],
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes over tools/trace_forward.py (one PC2 forward, B=16, "
                 "N=4096); FETCH_SIZE doubled (MI355X_MICROARCH.md: gfx950 tallies 128-byte read requests at 64 bytes); dispatches "
                 "attributed to ABI calls by launch order: one entry per (function, argument shape)",
       "kernels": {}, "unmatched": {}}
by_prefix = collections.defaultdict(list)
for fn, (prefix, _) in KERNEL_OF.items():
    by_prefix[prefix].append(fn)
for prefix, fns in by_prefix.items():
    d = [r for r in disp if r["kernel"].startswith(prefix)]
    calls = [(fn, args) for fn, args in log if fn in fns]
    if not d and not calls:
        continue
    if len(d) != len(calls):
        out["unmatched"][prefix] = {"dispatches": len(d), "abi_calls": len(calls)}
        continue
    acc = collections.OrderedDict()
    for r, (fn, args) in zip(d, calls):
        key = f"{fn}{tuple(args)}"      # = bench.py's f"{function}{tuple(shape)}"
        e = acc.setdefault(key, {"rocprof_kernel": r["kernel"], "grids": set(), "n": 0, "fetch": 0.0, "write": 0.0, "fn": fn, "args": args})
        e["grids"].add(int(r["grid_threads"])); e["n"] += 1
        e["fetch"] += float(r["fetch_mb_corrected_x2"]); e["write"] += float(r["write_mb"])
    for key, e in acc.items():
        alg = KERNEL_OF[e["fn"]][1]
        fm, wm = e["fetch"] / e["n"], e["write"] / e["n"]
        out["kernels"][key] = {"rocprof_kernel": e["rocprof_kernel"], "grid_threads": sorted(e["grids"]), "launches_in_forward": e["n"],
                               "fetch_mb": round(fm, 2), "write_mb": round(wm, 2), "bytes_per_launch": (fm + wm) * 2 ** 20,
                               "algorithmic_bytes_per_launch": alg(e["args"]) if alg else None}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(f"{len(out['kernels'])} (function, shape) entries, {len(out['unmatched'])} unmatched kernel families -> {sys.argv[4]}")
