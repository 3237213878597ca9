#!/usr/bin/env python3
"""Memory-side counters of one configuration of tools/config_rates.py (L1 / L2 request latency, address
translation, SQ wait kinds):  python tools/pmc_memory.py INDEX [--lib PATH]"""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from raytracer_amd import profiling
idx = sys.argv[1]
if "--lib" in sys.argv:
    os.environ["RTAMD_LIB"] = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
only = sys.argv[sys.argv.index("--passes") + 1].split(",") if "--passes" in sys.argv else None
passes = (
    ("lat", ("TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_READ_REQ_sum", "TCP_TCP_LATENCY_sum", "TCP_TOTAL_CACHE_ACCESSES_sum")),
    ("tlb", ("TCP_UTCL1_REQUEST_sum", "TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_TRANSLATION_HIT_sum", "TCP_PENDING_STALL_CYCLES_sum")),
    ("sq", ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS",
            "SQ_ACTIVE_INST_SCA", "SQ_INSTS_LDS")),
    ("icache", ("SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQC_ICACHE_MISSES_DUPLICATE", "SQC_ICACHE_BUSY_CYCLES",
                "SQC_DCACHE_REQ", "SQC_DCACHE_HITS", "SQC_DCACHE_MISSES")),
    ("levels", ("SQ_IFETCH_LEVEL", "SQ_INST_LEVEL_VMEM", "SQ_INST_LEVEL_SMEM", "SQ_INST_LEVEL_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES",
                "SQ_WAVES", "SQ_IFETCH")),
    ("sq2", ("SQ_INSTS_SMEM", "SQ_INSTS_BRANCH", "SQ_INSTS_FLAT", "SQ_WAIT_IFETCH", "SQ_IFETCH", "SQ_ACTIVE_INST_VMEM",
             "SQ_INST_CYCLES_SALU", "SQ_ACTIVE_INST_MISC")),
)
cmd = [sys.executable, os.path.join(REPO, "tools", "config_rates.py"), "--child", idx, "--frames", "1"]
out = {}
for p in passes:
    if only and p[0] not in only:
        continue
    got = profiling.collect(cmd, passes=(p,), log=lambda m: print(m, file=sys.stderr))
    if got:
        out.update(got)
print(json.dumps(out, indent=1, sort_keys=True))
