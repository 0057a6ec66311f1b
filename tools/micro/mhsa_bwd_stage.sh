# all GPU checks of the attention backward + its timings: bash tools/micro/mhsa_bwd_stage.sh   (through gpurun)
python -m pytest tests/test_gpu_training.py -q -x 2>&1 | tail -2
python -m pytest tests/test_gpu_parity.py -q -x -k "mhsa or transformer or attention" 2>&1 | tail -2
python tools/mhsa_bwd_bench.py 2>&1 | tail -2
bash tools/micro/mhsa_bwd_prof.sh 2>&1 | head -6 | tail -5
bash tools/micro/mhsa_bwd_pmc.sh > gpurun_out/mhsa_bwd_pmc.txt 2>&1
python tools/train_step_bench.py cfg3 3 Transformer 2>&1 | tail -2
python tools/train_step_bench.py cfg2 5 Transformer 2>&1 | tail -2
