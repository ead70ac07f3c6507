"""From a pmc_summary.py summary of the headline bench: the two small JSON files bench.py reads from profiles/
(traffic.json: fabric bytes per launch of the dominant kernels; pmc_counts.json: VALU wave-instructions per pair-dimension
of the pair kernels).  python tools/make_profile_json.py <summary.json> <tag> <outdir>"""
import json, sys
s = json.load(open(sys.argv[1])); tag = sys.argv[2]; out = sys.argv[3]
cpl = s["counters_per_launch"]
N, M, D = 1 << 20, 1024, 16
pair_dims_waves = N * M * D / 64.0
def find(sub):
    ks = [k for k in cpl if sub in k]
    return cpl[ks[0]] if ks else {}
gram, syrk, bwd, gemm = find("gram_kernel<"), find("syrk_kernel<"), find("gram_bwd_fast_kernel<"), find("gemm128_nt_kernel")
trsm = find("trsm_fused_kernel")
gcrt, scrt, gemm8 = find("gram_crt_kernel<"), find("crt_syrk_i8_deep_kernel"), find("crt_gemm_i8_kernel")
src = f"profiles/{tag}_pmc_summary.json (tools/profile_headline.sh {tag})"
traffic = {"headline": {"syrk": syrk.get("FETCH_SIZE_bytes", 0) + syrk.get("WRITE_SIZE_bytes", 0),
                        "gram": gram.get("FETCH_SIZE_bytes", 0) + gram.get("WRITE_SIZE_bytes", 0),
                        "bwd_gemm": gemm.get("FETCH_SIZE_bytes", 0) + gemm.get("WRITE_SIZE_bytes", 0),
                        "bwd_gram": bwd.get("FETCH_SIZE_bytes", 0) + bwd.get("WRITE_SIZE_bytes", 0),
                        "trsm": trsm.get("FETCH_SIZE_bytes", 0) + trsm.get("WRITE_SIZE_bytes", 0),
                        "crt_syrk": scrt.get("FETCH_SIZE_bytes", 0) + scrt.get("WRITE_SIZE_bytes", 0),
                        "gram_crt": gcrt.get("FETCH_SIZE_bytes", 0) + gcrt.get("WRITE_SIZE_bytes", 0),
                        "crt_gemm": gemm8.get("FETCH_SIZE_bytes", 0) + gemm8.get("WRITE_SIZE_bytes", 0)},
           "_note": "bytes per launch at the L2<->fabric boundary (TCC_EA requests; Infinity-Cache hits are included), from separate "
                    "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies "
                    "128-B requests at 64 B)",
           "_source": src}
counts = {"headline": {"gram_valu_wave_instr_per_pair_dim": gram.get("SQ_INSTS_VALU", 0) / pair_dims_waves,
                       "bwd_gram_valu_wave_instr_per_pair_dim": bwd.get("SQ_INSTS_VALU", 0) / pair_dims_waves,
                       "gram_crt_valu_wave_instr_per_pair_dim": gcrt.get("SQ_INSTS_VALU", 0) / pair_dims_waves},
          "_note": "SQ_INSTS_VALU per launch / (N*M*D/64) at the headline size; a property of the compiled kernel",
          "_source": src}
for k in ("gemm128_nt_kernel", "syrk_kernel<"):
    c = find(k)
    if "TCC_HIT_sum" in c:
        counts["headline"][k.strip("<") + "_L2_hit_rate"] = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0), 1)
json.dump(traffic, open(out + "/traffic.json", "w"), indent=1)
json.dump(counts, open(out + "/pmc_counts.json", "w"), indent=1)
print(json.dumps(counts, indent=1))
