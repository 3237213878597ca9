"""rocprofv3 counter passes around a child process, and the ceilings they are read against.

`collect()` runs a command under `rocprofv3 --pmc`, one run per pass (rocprofv3 does not split
counters over passes; gfx950 holds 8 SQ, 4 TCC - FETCH_SIZE uses 3, WRITE_SIZE 2 - and 2 GRBM
counters per run), and returns the per-dispatch average of every counter for the kernels
whose name contains `kernel`.  The caller must not have touched the GPU: the profiled
program is started as a fresh child.

Units (MI355X, ROCm 7.2): FETCH_SIZE and WRITE_SIZE are KiB at the L2's memory side;
FETCH_SIZE counts a 128-byte read request as 64 bytes and is doubled here; SQ_WAVE_CYCLES,
SQ_WAIT_* and SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_BUSY_CYCLES is summed
over the 32 shader engines; GRBM_GUI_ACTIVE over the 8 XCDs.
"""
import csv
import glob
import os
import shutil
import subprocess
import tempfile

HBM_PEAK_GBS = 8000.0       # HBM3E, MI355X data sheet
L2_PEAK_GBS = 34500.0       # aggregate of the eight 4 MiB L2s, measured read rate
N_SIMDS = 1024              # 256 CUs x 4
N_XCDS = 8

PASSES = (
    ("fetch", ("FETCH_SIZE",)),
    ("write", ("WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum")),
    ("sq", ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY",
            "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU", "GRBM_GUI_ACTIVE")),
)


def rocprofv3_path():
    return shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)


def collect(command, kernel=("trace_kernel", "refill_kernel"), passes=PASSES, timeout_s=240, log=None):
    """{counter: average per dispatch, "dispatches": n, "vgpr"/"accum_vgpr"/"sgpr"/"scratch"/"lds": ...}
    or None when rocprofv3 is missing or a pass fails."""
    exe = rocprofv3_path()
    if exe is None:
        return None
    out = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for name, counters in passes:
        work = tempfile.mkdtemp(prefix="rtamd_pmc_%s_" % name, dir="/tmp")
        try:
            cmd = ["timeout", "-s", "KILL", str(timeout_s), exe, "--pmc"] + list(counters) + \
                  ["--output-format", "csv", "-d", work, "--"] + list(command)
            try:
                p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                   timeout=timeout_s + 30)
            except (OSError, subprocess.TimeoutExpired) as e:
                if log:
                    log("pmc pass %s did not run: %s" % (name, e))
                return None
            if p.returncode != 0:
                if log:
                    log("pmc pass %s failed (%d): %s" % (name, p.returncode, p.stdout.decode(errors="replace")[-400:]))
                return None
            sums, n = {}, {}
            for path in glob.glob(os.path.join(work, "**", "*counter_collection.csv"), recursive=True):
                with open(path) as f:
                    for row in csv.DictReader(f):
                        if not any(k in row["Kernel_Name"] for k in ((kernel,) if isinstance(kernel, str) else kernel)):
                            continue
                        c = row["Counter_Name"]
                        sums[c] = sums.get(c, 0.0) + float(row["Counter_Value"])
                        n[c] = n.get(c, 0) + 1
                        for field, key in (("VGPR_Count", "vgpr"), ("Accum_VGPR_Count", "accum_vgpr"),
                                           ("SGPR_Count", "sgpr"), ("Scratch_Size", "scratch"), ("LDS_Block_Size", "lds")):
                            if field in row:
                                out[key] = int(float(row[field]))
                        out["kernel_name"] = row["Kernel_Name"]
            if not sums:
                if log:
                    log("pmc pass %s recorded no dispatch of %s" % (name, kernel))
                return None
            for c in sums:
                out[c] = sums[c] / n[c]
                out["dispatches"] = n[c]
        finally:
            shutil.rmtree(work, ignore_errors=True)
    return out


def ceilings(pmc, kernel_ms, executed_bytes=None):
    """Fractions of the HBM, L2 and vector-ALU ceilings for one launch that took kernel_ms.

    hbm: bytes the L2s exchanged with memory (FETCH_SIZE x 2 + WRITE_SIZE, KiB) / time / 8 TB/s.
    l2: bytes the kernel's loads asked for (counted by the kernel itself) / time / 34.5 TB/s -
        an upper bound on L2 traffic, since part of it is served by the CUs' L1s.
    valu: share of SIMD cycles in which one of the SIMD's waves was executing a vector-ALU
        instruction: SQ_ACTIVE_INST_VALU x 4 / (cycles per XCD x 1024 SIMDs); lane_utilisation =
        SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64): active lanes per vector instruction."""
    t = kernel_ms * 1e-3
    res = {}
    if pmc and "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        fetch, write = pmc["FETCH_SIZE"] * 1024.0 * 2.0, pmc["WRITE_SIZE"] * 1024.0
        res["hbm"] = {"bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write,
                      "achieved_GBs": (fetch + write) / t / 1e9, "peak_GBs": HBM_PEAK_GBS,
                      "frac": (fetch + write) / t / 1e9 / HBM_PEAK_GBS}
        if pmc.get("TCC_HIT_sum") is not None and pmc.get("TCC_MISS_sum") is not None:
            res["hbm"]["l2_hit_rate"] = pmc["TCC_HIT_sum"] / max(pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"], 1.0)
    if executed_bytes is not None:
        res["l2"] = {"bytes_per_launch": executed_bytes, "achieved_GBs": executed_bytes / t / 1e9,
                     "peak_GBs": L2_PEAK_GBS, "frac": executed_bytes / t / 1e9 / L2_PEAK_GBS}
    if pmc and "SQ_ACTIVE_INST_VALU" in pmc and "GRBM_GUI_ACTIVE" in pmc:
        simd_cycles = pmc["GRBM_GUI_ACTIVE"] / N_XCDS * N_SIMDS          # of the profiled dispatch itself
        busy = pmc["SQ_ACTIVE_INST_VALU"] * 4.0 / simd_cycles
        res["valu"] = {"insts_per_launch": pmc.get("SQ_INSTS_VALU"), "frac": busy,
                       "lane_utilisation": pmc["SQ_THREAD_CYCLES_VALU"] / (pmc["SQ_ACTIVE_INST_VALU"] * 64.0),
                       "wave_time_waiting": pmc["SQ_WAIT_ANY"] / pmc["SQ_WAVE_CYCLES"],
                       "vmem_insts_per_launch": pmc.get("SQ_INSTS_VMEM_RD", 0) + pmc.get("SQ_INSTS_VMEM_WR", 0)}
    if res:
        res["binding"] = max((k for k in res), key=lambda k: res[k]["frac"])
    return res
