G1="VALUBusy VALUUtilization SALUBusy"
G2="MemUnitStalled LdsUtil"
G3="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
G4="TCP_PENDING_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
G5="TA_BUSY_avr TCC_BUSY_avr TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum"
for w in parquet_sigma4_taylor2 gv_sigma5 parquet_sigma4; do tools/prof_counters.sh r3c_$w $w "$G1" "$G2" "$G3" "$G4" "$G5" > gpurun_out/r3c_$w.txt 2>&1; done
LAYOUT=sample_major tools/prof_counters.sh r3c_rm_parquet_sigma4 parquet_sigma4 "$G1" "$G2" "$G3" "$G4" "$G5" > gpurun_out/r3c_rm_parquet_sigma4.txt 2>&1
