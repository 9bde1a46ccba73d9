"""Per-GEMM MFMA utilisation from the two PMC passes of tools/pmc_mfma.sh -> profiles/<round>/pmc_mfma.txt"""
import csv, glob, re, sys
from collections import defaultdict
dst = sys.argv[1] if len(sys.argv) > 1 else "profiles/r1/pmc_mfma.txt"
out = ["# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE (one pass per variant),",
       "# bench.py --steps 2 --warmup 1, products-shaped.  GRBM_GUI_ACTIVE is summed over the 8 XCDs;",
       "# MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)  (v_mfma_f32_32x32x2_f32 = 64 busy cycles)",
       "# the default bench command also runs the hidden layer in f32 (split3_*: three-way bf16 split, six bf16 products) and in bf16 (gemm_bf16_*, dw_bf16_*)"]
for v, label in (("auto", "input-side aggregation path (gemm_kreg<2,13,false,false>: out = Z W^T | gemm_kreg<3,8,true,true>: q = g W | inagg_dw_stream<2,3,true>: dW = g^T Z)"),
                 ("off", "projected-row path (gemm_f32_v2<false,false,false>: h = x W^T | gemm_f32_v2<true,true,true>: dW = d_h^T x)")):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(f"gpurun_out/pmc_mfma_{v}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            m = re.search(r"((?:gemm_\w+|split3_\w+|dw_bf16\w*|inagg_dw_stream_kernel)<[^>]*>)", row["Kernel_Name"])
            if m:
                acc[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out.append(f"== {label}")
    for k in sorted(acc):
        c = {n: sum(x) / len(x) for n, x in acc[k].items()}
        util = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)
        out.append(f"{k:44s} n={len(acc[k]['GRBM_GUI_ACTIVE'])} GRBM_GUI_ACTIVE {c['GRBM_GUI_ACTIVE']:.0f}  MFMA_BUSY {c['SQ_VALU_MFMA_BUSY_CYCLES']:.0f}  "
                   f"MOPS_F32 {c['SQ_INSTS_VALU_MFMA_MOPS_F32']:.0f}  MOPS_BF16 {c.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', 0):.0f}  SQ_BUSY {c['SQ_BUSY_CYCLES']:.0f}  -> MFMA utilisation {100 * util:.1f} %")
open(dst, "w").write("\n".join(out) + "\n")
print("\n".join(out))
