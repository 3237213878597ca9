#!/usr/bin/env python3
"""Extra SQ / TA counters for one configuration of tools/config_rates.py:
  python tools/pmc_config.py INDEX [--lib PATH]"""
import json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from raytracer_amd import profiling
idx = sys.argv[1]
if "--lib" in sys.argv:
    os.environ["RTAMD_LIB"] = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
passes = (
    ("a", ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
           "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM", "GRBM_GUI_ACTIVE")),
    ("b", ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS",
           "SQ_INST_CYCLES_SMEM", "SQ_WAVES")),
    ("c", ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64",
           "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_FLAT", "SQ_INST_CYCLES_SALU")),
    ("d", ("TA_TA_BUSY_sum", "TA_FLAT_READ_WAVEFRONTS_sum", "TCC_HIT_sum", "TCC_MISS_sum")),
)
cmd = [sys.executable, os.path.join(REPO, "tools", "config_rates.py"), "--child", idx, "--frames", "1"]
out = profiling.collect(cmd, passes=passes, log=lambda m: print(m, file=sys.stderr))
print(json.dumps(out, indent=1, sort_keys=True))
