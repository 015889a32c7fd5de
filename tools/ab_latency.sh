#!/bin/bash
# single-proof latency under scheduling variants (stream priorities, G1 CU share)
run() { echo -n "$1: "; shift; env REPS=12 "$@" timeout -k 10 200 python tools/latency_trace.py 2>/dev/null | tail -1; }
run default
run lanes_high ZKDL_LANE_PRIO=1
run g1_low ZKDL_G1_PRIO=-1
run lanes_high_g1_low ZKDL_LANE_PRIO=1 ZKDL_G1_PRIO=-1
run g1_33 ZKDL_G1_CU_PERCENT=33
run g1_75 ZKDL_G1_CU_PERCENT=75
run derive DERIVE=1
