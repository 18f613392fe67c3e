#!/bin/bash
# Round 5, verdict item 1: which rocprofv3 counter passes give the HBM-side traffic of the wave kernel, how long each takes, and whether the
# per-candidate instruction counts of a 6,144-replica launch carry over to the timed 24,576-replica launch.  Runs on the GPU box.
# usage: scripts/r05_pmc_probe.sh    (writes gpurun_out/r05_pmc/*)
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r05_pmc; mkdir -p $O; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_[A-Z0-9_]*\|TCC_REQ[A-Z0-9_]*\|TCC_HIT[A-Z0-9_]*\|TCC_MISS[A-Z0-9_]*\|TCP_TCC_[A-Z0-9_]*" | sort -u > $O/tcc_counters.txt
CH="python $R/bench.py --gpus 1 --solve-seconds 0 --no-cpu-baseline --no-pmc --pmc-child"
pass() {  # name replicas steps ls_steps deadline counters...
  name=$1; rep=$2; st=$3; ls=$4; dl=$5; shift 5
  d=/tmp/r05_$name/pmc_0; rm -rf /tmp/r05_$name
  t0=$(date +%s.%N)
  (cd /tmp && timeout -s KILL $dl rocprofv3 --pmc "$@" -f csv -d $d -o b -- $CH --steps $st --warmup 2 --ls-steps $ls --replicas $rep --pmc-child-out $O/$name.work.json > $O/$name.log 2>&1)
  rc=$?
  t1=$(date +%s.%N)
  python $R/scripts/pmc_dump.py /tmp/r05_$name k_list_search_wave > $O/$name.json 2>/dev/null
  echo "$name rc=$rc seconds=$(echo "$t1 - $t0" | bc) counters=$*" | tee -a $O/summary.txt
}
pass ea_rdwr_short   6144 4 50 120 TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
pass ea_rd32_short   6144 4 50 120 TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum
pass fetch_short     6144 4 50 120 FETCH_SIZE
pass write_short     6144 4 50 120 WRITE_SIZE
pass fetch_full      6144 20 200 150 FETCH_SIZE
pass insts_6144      6144 4 200 120 SQ_INSTS_VALU SQ_INSTS_SALU
pass insts_24576a    24576 4 200 120 SQ_INSTS_VALU SQ_INSTS_SALU
pass insts_24576b    24576 4 200 120 SQ_INSTS_VALU SQ_INSTS_SALU
pass insts_24576c    24576 4 200 120 SQ_INSTS_VALU SQ_INSTS_SALU
pass cyc_6144        6144 4 200 120 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT
pass cyc_9216        9216 4 200 120 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT
pass cyc_12288       12288 4 200 120 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass cyc_24576       24576 4 200 120 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
cat $O/summary.txt
