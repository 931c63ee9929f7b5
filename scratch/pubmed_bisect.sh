#!/bin/bash
run() { echo "== $1"; env $2 timeout -k 10 200 python bench.py --config ${CFG:-pubmed} --steps 200 --warmup 20 --cpu-baseline-steps 0 --no-roofline 2>&1 | tail -1 | cut -c1-200; }
run default ""
run nosplit "BLISS_SPLIT_FORWARD=0"
run nomfma "BLISS_SAGE_MFMA=0"
run nofusedce "BLISS_FUSED_CE=0"
cp bliss_gnn_amd/libbliss_gnn.so /tmp/base.so; cp scratch/variants/libbliss_oldshapes.so bliss_gnn_amd/libbliss_gnn.so
run oldshapes ""
cp /tmp/base.so bliss_gnn_amd/libbliss_gnn.so
