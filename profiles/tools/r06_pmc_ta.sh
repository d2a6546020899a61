#!/bin/bash
# round 6: the CU's vector-memory path under k_seg_fb (TA / TCP counters), one rocprofv3 --pmc pass per group, each under its own time limit
# (a first attempt with seven TA counters in one pass hung rocprofv3 for the whole 20-minute lease)
set -u
cd "$(dirname "$0")/../.."
pass() { timeout 150 bash profiles/pmc_pass.sh "$@" || echo "pass $1: timed out / failed"; }
pass gpurun_out/r06_pmc_ta_a.json TA_TA_BUSY_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum
pass gpurun_out/r06_pmc_ta_b.json TA_FLAT_WRITE_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum
pass gpurun_out/r06_pmc_ta_c.json TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
pass gpurun_out/r06_pmc_ta_d.json TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
pass gpurun_out/r06_pmc_ta_e.json TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass gpurun_out/r06_pmc_ta_f.json TCP_GATE_EN1_sum TCP_GATE_EN2_sum
