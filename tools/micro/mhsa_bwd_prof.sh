# per-kernel times of the attention backward at the config-3 shape: bash tools/micro/mhsa_bwd_prof.sh   (through gpurun)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/mhsa_bwd
rm -rf /tmp/kb
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kb -o kt -- python3 $R/tools/mhsa_bwd_bench.py 40962,16,64 > $R/gpurun_out/mhsa_bwd/log.txt 2>&1
python3 $R/tools/summarize_trace.py /tmp/kb > $R/gpurun_out/mhsa_bwd/summary.txt 2>&1
tail -1 $R/gpurun_out/mhsa_bwd/log.txt
head -12 $R/gpurun_out/mhsa_bwd/summary.txt | cut -c1-60,100-170
