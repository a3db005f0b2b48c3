"""Per-kernel-family roofline table (markdown) from a serialized kernel trace and the two PMC summaries of the same tree.
usage: python probes/roofline_table.py <kernel_stats.txt> <pmc_FETCH_SIZE.txt> <pmc_WRITE_SIZE.txt> <steps in trace> <steps in pmc> <config>
FLOPs: the algorithmic GEMM FLOPs of the configuration (bench.py's per-launch 2MNK accounting; NT / TN split as measured by it),
attention 4*B*H*Lq*Lk*64 forward x 3.5 for forward + backward.  Bytes: measured (PMC, FETCH_SIZE doubled per MI355X_MICROARCH.md).
Roofs: 2500 TFLOP/s dense bf16 MFMA, 8.0 TB/s HBM (6.3 achievable)."""
import re
import sys

FLOPS = {   # TFLOP per step: (NT algorithmic, TN algorithmic, attention)
    # NT / TN: launches x GFLOP per launch of bench.py's accounting (profiles/r03k_bench_*.json); attention: 3.5 x the forward's
    # 4*B*H*Lq*Lk*64 over vision (+ text / fusion self- and cross-attention for base)
    "base": (7.556, 3.786, 0.399), "large": (25.825, 12.927, 5.62), "video": (4.706, 2.360, 0.40),
}


def family(n):
    if n.startswith("gemm_nt"):
        return "NT GEMM (forward linears, input gradients)"
    if n.startswith("gemm_tn"):
        return "TN GEMM (weight gradients)"
    if n.startswith("attn_"):
        return "attention fwd / dQ / dK,dV"
    if n.startswith("layernorm"):
        return "LayerNorm fwd / bwd"
    if n.startswith("reduce_partials") or n.startswith("colsum") or n.startswith("layerscale") or n.startswith("rowscale"):
        return "column sums, layer-scale bwd, stage-2 reductions"
    if n.startswith("relpos"):
        return "rel-pos bias gather / gradient"
    if n.startswith("cast_") or n.startswith("copy_f32"):
        return "bf16 weight casts"
    if "at::" in n or n.startswith("__amd_rocclr") or "elementwise" in n or "cub::" in n or "rocprim" in n:
        return "torch-native elementwise / copies / fills"
    return "heads, embeddings, CE, sampling"


def table(path, ncols):
    """[(name, calls, first value column)]; names may contain spaces (numeric columns are counted from the right) and may
    repeat (truncated template names)."""
    rows = []
    for line in open(path):
        if line.startswith("kernel") or line.startswith("#"):
            continue
        parts = line.split()
        if len(parts) <= ncols:
            continue
        try:
            nums = [float(x) for x in parts[-ncols:]]
        except ValueError:
            continue
        rows.append((" ".join(parts[:-ncols]), int(nums[0]), nums[1]))
    return rows


stats = table(sys.argv[1], 6)
have_pmc = sys.argv[2] != "-"
fetch, write = (table(sys.argv[2], 4), table(sys.argv[3], 4)) if have_pmc else ([], [])
steps_t, steps_p, cfg = int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
fam = {}
for n, calls, total_us in stats:
    f = fam.setdefault(family(n), dict(calls=0, us=0.0, rd=0.0, wr=0.0))
    f["calls"] += calls
    f["us"] += total_us
for n, calls, avg_kb in fetch:
    fam.setdefault(family(n), dict(calls=0, us=0.0, rd=0.0, wr=0.0))["rd"] += calls * avg_kb * 1024 * 2
for n, calls, avg_kb in write:
    fam.setdefault(family(n), dict(calls=0, us=0.0, rd=0.0, wr=0.0))["wr"] += calls * avg_kb * 1024
tot = sum(f["us"] for f in fam.values())
fl = dict(zip(("NT GEMM (forward linears, input gradients)", "TN GEMM (weight gradients)", "attention fwd / dQ / dK,dV"), FLOPS[cfg]))
print("| kernel family | launches / step | ms / step | share | algorithmic TFLOP / step | achieved TFLOP/s | of 2.5 PF | HBM GB / step (PMC) | achieved TB/s | of 8 TB/s (of 6.3 achievable) |")
print("|---|---|---|---|---|---|---|---|---|---|")
for name, f in sorted(fam.items(), key=lambda kv: -kv[1]["us"]):
    ms = f["us"] / steps_t / 1e3
    gb = (f["rd"] + f["wr"]) / steps_p / 1e9
    tf = fl.get(name)
    mem = "%.2f | %.2f | %.0f %% (%.0f %%)" % (gb, gb / ms, 100 * gb / ms / 8.0, 100 * gb / ms / 6.3) if have_pmc else "- | - | -"
    print("| %s | %.0f | %.2f | %.1f %% | %s | %s | %s | %s |" % (
        name, f["calls"] / steps_t, ms, 100 * f["us"] / tot, "%.2f" % tf if tf else "-", "%.0f" % (tf / ms * 1e3) if tf else "-",
        "%.1f %%" % (100 * tf / ms * 1e3 / 2500) if tf else "-", mem))
print("| **all kernels, serialized** | %.0f | **%.2f** | | | | | | | |" % (sum(f["calls"] for f in fam.values()) / steps_t, tot / steps_t / 1e3))
