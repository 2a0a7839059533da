import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f))
    print("==", f)
    for k in ["_kernel_ms_profiled_mean", "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES",
              "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS",
              "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "l2_hit_rate", "l1_hit_rate_est", "SQ_LDS_BANK_CONFLICT",
              "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_FMA_F32", "GRBM_GUI_ACTIVE", "valu_active_frac_of_wave_cycles",
              "wait_any_frac_of_wave_cycles", "SQ_WAVES", "FETCH_SIZE", "WRITE_SIZE"]:
        print("  %-34s %s" % (k, d.get(k)))
