"""profiles/pmc_traffic.json from the two PMC summaries (probes/pmc_summary.py output) of the same bench command.
usage: python probes/pmc_traffic.py <FETCH_SIZE.txt> <WRITE_SIZE.txt> <tag> > pmc_traffic.json
Units: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB; FETCH_SIZE is doubled (gfx950 tallies 128-B read requests at 64 B,
MI355X_MICROARCH.md HBM section).  The correction is cross-checked on gemm_tn256_reduce_kernel when it appears."""
import json
import re
import sys


def table(path):
    rows = {}
    for line in open(path):
        m = re.match(r"^(\S.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
        if m and not line.startswith("kernel"):
            rows[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)))
    return rows


fetch, write, tag = table(sys.argv[1]), table(sys.argv[2]), sys.argv[3]
# the NT family: 128-column kernels, the 256-column kernel; the MLM head's epilogue variants 8 / 9 are NT launches as well
names = [n for n in fetch if n.startswith("gemm_nt")]          # gemm_nt_kernel, gemm_nt256_kernel, gemm_nt256s3_kernel (sum_slices: negligible)
calls = sum(fetch[n][0] for n in names)
fb = sum(fetch[n][0] * fetch[n][1] for n in names) / calls * 1024 * 2
wb = sum(write[n][0] * write[n][1] for n in names if n in write) / max(sum(write[n][0] for n in names if n in write), 1) * 1024
# whole step: every kernel's fetch (x 2) + write bytes, per profiled step (patchify_kernel runs once per step)
steps = max(fetch.get("patchify_kernel", (1, 0))[0], 1)
total = (sum(c * a for c, a in fetch.values()) * 2 + sum(c * a for c, a in write.values())) * 1024 / steps
out = {"kernel": "gemm_nt_kernel<*> + gemm_nt256_kernel<*> + gemm_nt256s3_kernel<*>", "launches": calls, "fetch_bytes_per_launch": int(fb), "write_bytes_per_launch": int(wb),
       "hbm_bytes_per_launch": int(fb + wb),
       "whole_step_hbm_gb": round(total / 1e9, 2), "whole_step_steps_profiled": steps,
       "whole_step_hbm_floor_ms": {"at_6.3_TB/s_achievable": round(total / 6.3e9, 2), "at_8_TB/s_peak": round(total / 8e9, 2)},
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (probes/run_pmc.sh) over "
                 "`bench.py --serialize --no-graph --steps 2 --warmup 1` (base config); KB units; FETCH_SIZE doubled (gfx950 counts "
                 "128-B read requests at 64 B: MI355X_MICROARCH.md, HBM section)",
       "files": ["profiles/%s_pmc_FETCH_SIZE.txt" % tag, "profiles/%s_pmc_WRITE_SIZE.txt" % tag]}
print(json.dumps(out, indent=1))
