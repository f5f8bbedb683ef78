#!/bin/bash
out=gpurun_out/r04ss; mkdir -p $out
O=$GRAFT_REPO_ROOT/superscreen_amd/lib/libssa_oldasm.so
for rep in 1 2; do
SSA_LIB_PATH=$O timeout 600 python tools/r04/asm_timing.py 91 129 > $out/old_$rep.txt 2>&1
timeout 600 python tools/r04/asm_timing.py 91 129 > $out/new_$rep.txt 2>&1
done
paste -d'\n' $out/old_1.txt $out/new_1.txt | grep K=; echo; paste -d'\n' $out/old_2.txt $out/new_2.txt | grep K=
timeout 900 python -m pytest tests -m gpu -x -q --timeout 600 -k "assemble or system or golden or fixture or solve_film" 2>&1 | tail -2
