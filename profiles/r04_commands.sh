# The round-4 profile set: ONE gpurun call on the final tree (python profiles/summarize.py gpurun_out/r4final r04 afterwards).
# Default bench = 2 clips per GPU (bench.py --clips); the traces and the LDS pass run --clips 1 so that per-kernel durations compare with r03.
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4final
mkdir -p $O
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -1 $O/bench_default.json | cut -c1-600 &&
timeout -k 10 250 python bench.py --no-cpu-baseline --clips 1 > $O/bench_1clip.json 2> $O/bench_1clip.err &&
VSRLAB_AMD_CHAIN=0 timeout -k 10 250 python bench.py --no-cpu-baseline --clips 1 > $O/bench_chain_off.json 2> $O/bench_chain_off.err &&
timeout -k 10 250 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_torchrun_1rank.json 2> $O/bench_torchrun_1rank.err &&
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace2s -o bench -- python3 bench.py --clips 1 --steps 1 --warmup 1 --no-cpu-baseline > $O/trace2s.log 2>&1 &&
VSRLAB_AMD_SINGLE_STREAM=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace1s -o bench -- python3 bench.py --clips 1 --steps 1 --warmup 1 --no-cpu-baseline > $O/trace1s.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o rl -- python3 bench.py --roofline-only > $O/pmc_fetch.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o rl -- python3 bench.py --roofline-only > $O/pmc_write.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o rl -- python3 bench.py --roofline-only > $O/pmc_mfma.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_step_fetch -o st -- python3 bench.py --clips 1 --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_step_fetch.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_step_write -o st -- python3 bench.py --clips 1 --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_step_write.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_lds -o lds -- python3 bench.py --train-flow --clips 1 --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_lds.log 2>&1 &&
timeout -k 10 200 python tools/ab_chain.py vsrlab_amd/lib/libvsrlab_hip.so vsrlab_amd/lib/libvsrlab_hip_conv3x3_chain_abl0.so > $O/clock_chain.log 2>&1 &&
timeout -k 10 200 python tools/ab_conv.py 5 libvsrlab_hip.so libvsrlab_hip_conv3x3_persist_abl0.so > $O/clock.log 2>&1 &&
timeout -k 10 200 python tools/ab_wgrad.py 5 libvsrlab_hip.so libvsrlab_hip_wgrad_mfma_abl0.so libvsrlab_hip_wgrad_mfma_abl4.so > $O/clock_wgrad.log 2>&1 &&
timeout -k 10 250 python bench.py --no-cpu-baseline --clips 1 --arena diet > $O/bench_diet.json 2> $O/bench_diet.err &&
timeout -k 10 250 python bench.py --no-cpu-baseline --clips 1 --train-flow > $O/bench_train_flow.json 2> $O/bench_train_flow.err &&
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/gan -o gan -- python3 tools/bench_gan.py 1 > $O/gan_prof.log 2>&1 &&
timeout -k 10 200 python tools/bench_gan.py 3 > $O/gan.log 2>&1; tail -1 $O/gan.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
ls $O
