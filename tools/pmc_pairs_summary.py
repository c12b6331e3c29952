#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc passes of tools/pmc_pairs.sh into the JSON summaries committed under profiles/: per case the averaged counters per
launch of the kernel under study and the derived fractions (matrix pipe busy, waves parked / issue-stalled, L2 hit rate, LDS bank conflicts,
HBM bytes with the gfx950 FETCH_SIZE correction).  usage: pmc_pairs_summary.py <dir> <gemm.json> <attention.json>"""
import collections, csv, glob, json, os, sys

root, gemm_out, attn_out = sys.argv[1:4]
CUS = 256


def case(name, match):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, name, "g*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if match in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    c = {k: sum(v) / len(v) for k, v in acc.items()}
    d = {}
    if "SQ_WAVE_CYCLES" in c and c.get("SQ_WAVES"):
        kc = 4.0 * c["SQ_WAVE_CYCLES"] / c["SQ_WAVES"]            # SQ_WAVE_CYCLES counts quad-cycles
        d["kernel_cycles_per_wave(4*SQ_WAVE_CYCLES/SQ_WAVES)"] = round(kc)
        d["wave_cycles_parked_frac(SQ_WAIT_ANY/SQ_WAVE_CYCLES)"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3)
        d["issue_stalled_frac(SQ_WAIT_INST_ANY/SQ_WAVE_CYCLES)"] = round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            per_simd = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * CUS)  # counts cycles, summed over the SIMDs of the chip
            d["mfma_busy_cycles_per_simd"] = round(per_simd)
            # one 8-wave workgroup per CU at a time in both kernels: rounds = workgroups a CU runs one after the other (1 for the
            # persistent GEMM, 3 for the attention's 768 workgroups); the busy cycles are summed over them, the lifetime is one's
            rounds = max(1.0, c["SQ_WAVES"] / (8.0 * CUS))
            d["workgroup_rounds_per_cu"] = round(rounds, 2)
            d["matrix_pipe_busy_frac(of the workgroups' lifetimes)"] = round(per_simd / (kc * rounds), 3)
    if "TCC_HIT_sum" in c:
        d["l2_hit_rate"] = round(c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1.0), 3)
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_frac"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 3)
    if "SQ_INSTS_MFMA" in c:
        d["salu_per_mfma"] = round(c["SQ_INSTS_SALU"] / max(c["SQ_INSTS_MFMA"], 1.0), 2)
    if "FETCH_SIZE" in c:
        d["hbm_read_bytes(FETCH_SIZE*1024*2, gfx950 correction)"] = round(c["FETCH_SIZE"] * 1024 * 2)
    if "WRITE_SIZE" in c:
        d["hbm_write_bytes(WRITE_SIZE*1024)"] = round(c["WRITE_SIZE"] * 1024)
    if "GRBM_GUI_ACTIVE" in c:
        d["GRBM_GUI_ACTIVE_per_XCD"] = round(c["GRBM_GUI_ACTIVE"] / 8)   # the dispatch's duration in GPU clocks (sum over the 8 XCDs / 8)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            # against the whole dispatch (launch ramp and tail included; THE figure for a kernel whose workgroups come in several rounds,
            # where a wave's lifetime is a fraction of the dispatch: the attention kernel runs 3 rounds of 768 workgroups)
            d["matrix_pipe_busy_frac_of_dispatch(GRBM)"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * CUS) / (c["GRBM_GUI_ACTIVE"] / 8), 3)
    return {"counters": {k: round(v) for k, v in sorted(c.items())}, "derived": d}


note = ("rocprofv3 --pmc passes (tools/pmc_pairs.sh: one counter group per run, no trace domains) over tools/pairs_one.py; averages per launch of the "
        "SHIPPED kernel on random operands.  Profiled passes clock lower than un-profiled ones (MI355X_MICROARCH.md 'DVFS give-back' item 2): fractions, not times.")
g = {"note": note + "  gemm_pairs8s_kernel on the four ViT-S/16 block shapes (25216 rows): qkv 1152 x 384 pairs out, proj 384 x 384 fp32 + residual, "
             "fc1 1536 x 384 GELU -> pairs, fc2 384 x 1536 fp32 + residual."}
for name in ("qkv", "proj", "fc1", "fc2"):
    g[name] = case(name, "gemm_pairs8")
json.dump(g, open(gemm_out, "w"), indent=1)
a = {"note": note + "  attention_fwd_pairs_kernel, 128 frames x 197 tokens x 6 heads (one ViT-S/16 layer of the C2 step), pairs in / pairs out.",
     "attn_pairs": case("attn", "attention_fwd_pairs_kernel")}
if os.path.isdir(os.path.join(root, "attn785")):
    a["attn_pairs_kv_tiled_785_tokens(64 frames x 6 heads: a C5 layer)"] = case("attn785", "attention_fwd_pairs_flash_kernel")
if os.path.isdir(os.path.join(root, "tn")):
    g["wgrad_tn(6304 rows, dW 1536 x 384: gemm_pairs_tn_kernel, two 4-wave workgroups per CU - its busy fraction counts 8-wave rounds)"] = case("tn", "gemm_pairs_tn_kernel")
    json.dump(g, open(gemm_out, "w"), indent=1)
json.dump(a, open(attn_out, "w"), indent=1)
for k, v in list(g.items())[1:] + list(a.items())[1:]:
    print(k, v["derived"])
