"""profiles/rNN_pmc_dispatches.csv (tools/pmc_summary.py, per-dispatch form) + the ABI-call log of the same forward
(tools/trace_forward.py with BDM_ABI_LOG) -> profiles/rNN_pmc_traffic.json keyed like bench.py's roofline rows ("bdm_function(shape)").

    pmc_to_json.py <dispatches.csv> <abi_log.json> <commit> <out.json>

Every entry is ONE (function, argument shape): the dispatches of a kernel family are attributed to the ABI calls that launch it
BY ORDER (the i-th dispatch of `sparse_gemm_s3_kernel` in the marked forward belongs to the i-th bdm_sparse_conv_gemm_s3 call of the
log), so two shapes that share a (kernel, grid) never share an entry.  A function whose calls and dispatches do
not pair up one to one is reported under "unmatched" instead of being averaged."""
import collections, csv, json, sys

# Kernel FAMILIES: the rocprof kernel-name prefixes a group of C-ABI functions launches, how many dispatches of the family one call
# makes, and per function the algorithmic HBM bytes per CALL from its int arguments (None: data-dependent or not priced in bytes).
_conv = lambda a: 4 * a[0] * a[3] ** 3 * (a[1] + a[2])
_pw = lambda a: 4 * a[0] * a[3] * (a[1] + a[2])                      # (b, m, k, n): x read once, y written once
FAMILIES = [
    (("conv3d_h2q_kernel",), 1, {"bdm_conv3d_3x3x3_h2_gn": _conv, "bdm_conv3d_3x3x3_h2": _conv}),
    # compact first / second convolution over the dilated voxel lists: rows in, compact rows out -- the list lengths are data (None)
    (("sconv_dil_kernel",), 1, {"bdm_sparse_conv_dil_gn": None, "bdm_sparse_conv_dil": None, "bdm_sparse_conv_dil_h2_gn": None}),
    (("sparse_gemm_s3_kernel",), 1, {"bdm_sparse_conv_gemm_s3": None, "bdm_sparse_conv_gemm_s3_cb": None}),
    (("sparse_gemm_h2_kernel",), 1, {"bdm_sparse_conv_gemm_h2": None, "bdm_sparse_conv_gemm_h2_cb": None}),
    (("sparse_gather_v4_kernel",), 1, {"bdm_sparse_conv_gather_gn": lambda a: 4 * a[0] * a[1] * a[2] ** 3,
                                       "bdm_sparse_conv_gather": lambda a: 4 * a[0] * a[1] * a[2] ** 3}),
    (("sparse_rows_from_map_kernel",), 1, {"bdm_sparse_conv_rows_from_map": None}),
    (("attn_flash_h2_kernel",), 1, {"bdm_attention_core_h2": None}),
    (("ball_query_kernel",), 1, {"bdm_ball_query": None}),
    (("pw_gemm_kernel", "pw_skinny_kernel"), 1, {"bdm_pointwise_conv_gn": _pw, "bdm_pointwise_conv_gn_add": _pw, "bdm_pointwise_conv_gn_bb": _pw, "bdm_pointwise_conv": _pw}),
    # fused first set-abstraction level: row repack + three recompute passes per call; algorithmic = features + indices + output
    (("sa_rows_kernel", "sa_mlp2_kernel"), 4, {"bdm_sa_mlp2_fused": lambda a: 4 * a[0] * ((3 + a[1]) * a[2] + a[3] * a[4] + a[6] * a[3])}),
    (("to_h2_rows_kernel",), 1, {"bdm_group_norm_to_h2_rows": None}),
    (("se_rows_partial_kernel", "se_rows_fc_kernel"), 2, {"bdm_se_gate_gn_rows_pf": None, "bdm_se_gate_gn_rows": None}),
    (("devox_rows_kernel",), 1, {"bdm_devoxelize_gn_gate_add_rows_pf": None, "bdm_devoxelize_gn_gate_add_rows": None}),
    (("vox_dilate_kernel",), 1, {"bdm_voxel_dilate": None, "bdm_voxel_dilate_again": None}),
    # small-grid PVConv tail (+ the next PVConv's operand): grid in, points out
    (("pv_tail_small",), 1, {"bdm_pvconv_tail_small": lambda a: 4 * a[0] * a[1] * (a[3] ** 3 + 2 * a[2])}),
]
disp = list(csv.DictReader(open(sys.argv[1])))
log = json.load(open(sys.argv[2]))   # [[function, [int args...]], ...] in call order
out = {"commit": sys.argv[3],
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes over tools/trace_forward.py (one PC2 forward, B=16, "
                 "N=4096); FETCH_SIZE doubled (MI355X_MICROARCH.md: gfx950 tallies 128-byte read requests at 64 bytes); dispatches "
                 "attributed to ABI calls by launch order: one entry per (function, argument shape)",
       "kernels": {}, "unmatched": {}}
for prefixes, per_call, fns in FAMILIES:
    d = [r for r in disp if r["kernel"].startswith(prefixes)]
    calls = [(fn, args) for fn, args in log if fn in fns]
    if not d and not calls:
        continue
    if len(d) != per_call * len(calls):
        out["unmatched"]["+".join(prefixes)] = {"dispatches": len(d), "abi_calls": len(calls), "dispatches_per_call": per_call}
        continue
    acc = collections.OrderedDict()
    for i, (fn, args) in enumerate(calls):
        key = f"{fn}{tuple(args)}"      # = bench.py's f"{function}{tuple(shape)}"
        mine = d[i * per_call:(i + 1) * per_call]
        e = acc.setdefault(key, {"kernels": [], "grids": set(), "n": 0, "fetch": 0.0, "write": 0.0, "fn": fn, "args": args})
        for r in mine:
            if r["kernel"] not in e["kernels"]:
                e["kernels"].append(r["kernel"])
            e["grids"].add(int(r["grid_threads"]))
        e["n"] += 1
        e["fetch"] += sum(float(r["fetch_mb_corrected_x2"]) for r in mine); e["write"] += sum(float(r["write_mb"]) for r in mine)
    for key, e in acc.items():
        alg = fns[e["fn"]]
        fm, wm = e["fetch"] / e["n"], e["write"] / e["n"]
        out["kernels"][key] = {"rocprof_kernel": " + ".join(e["kernels"]), "grid_threads": sorted(e["grids"]), "launches_in_forward": e["n"],
                               "dispatches_per_call": per_call, "fetch_mb": round(fm, 2), "write_mb": round(wm, 2),
                               "bytes_per_launch": (fm + wm) * 2 ** 20, "algorithmic_bytes_per_launch": alg(e["args"]) if alg else None}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(f"{len(out['kernels'])} (function, shape) entries, {len(out['unmatched'])} unmatched kernel families -> {sys.argv[4]}")
