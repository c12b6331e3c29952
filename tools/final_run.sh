R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r2_final_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r2_final_c2.json 2> gpurun_out/r2_final_c2.err
python bench.py --steps 50 --warmup 10 --batch_size 2 --num_frames 2 --num_clusters 50 --no_alt_precision > gpurun_out/r2_final_c1.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --use_teacher --use_queue --queue_size 2048 --no_cpu_baseline --no_alt_precision > gpurun_out/r2_final_c3.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --architecture dino-b16 --num_frames 8 --num_clusters 400 --batch_size 16 --precision bf16 --no_alt_precision > gpurun_out/r2_final_c4.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --architecture dino-b16 --num_frames 8 --num_clusters 400 --batch_size 16 --no_alt_precision --no_cpu_baseline > gpurun_out/r2_final_c4_f32.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --architecture dino-s8 --batch_size 16 --no_alt_precision --no_cpu_baseline > gpurun_out/r2_final_c5.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -o c2 -- python3 $R/bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_alt_precision > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final_c4 -o c4 -- python3 $R/bench.py --steps 10 --warmup 3 --architecture dino-b16 --num_frames 8 --num_clusters 400 --batch_size 16 --precision bf16 --no_alt_precision --no_cpu_baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final_c1 -o c1 -- python3 $R/bench.py --steps 50 --warmup 10 --batch_size 2 --num_frames 2 --num_clusters 50 --no_alt_precision --no_cpu_baseline > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o f -- python3 $R/bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_alt_precision > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o w -- python3 $R/bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_alt_precision > /dev/null 2>&1
cd $R
find gpurun_out/prof_final gpurun_out/pmc_fetch gpurun_out/pmc_write -type f | head -30
find gpurun_out/prof_final gpurun_out/prof_final_c4 gpurun_out/prof_final_c1 -name "*kernel_trace.csv" -delete
find gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*kernel_trace.csv" -delete
du -sh gpurun_out/prof_final gpurun_out/pmc_fetch gpurun_out/pmc_write
cat gpurun_out/r2_final_tests.log
for f in c2 c1 c3 c4 c4_f32 c5; do echo == $f; cut -c1-900 gpurun_out/r2_final_$f.json; echo; done
