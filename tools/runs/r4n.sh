#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r4n_tests.log 2>&1; tail -3 gpurun_out/r4n_tests.log
for v in TGD TGE TGD TGE; do PTX_AB_LIBRARY=$PWD/.ab/lib$v.so python tools/gpu_c5_leg.py 72 2>/dev/null; done
AB_REPS=1 AB_SCENES=cornellSpaceship20k.txt bash tools/ab_c5.sh TGD TGE
