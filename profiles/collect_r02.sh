# Round-2 evidence, collected on the MI355X box (one rocprofv3 pass per counter set, as gpurun requires):
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash profiles/collect_r02.sh'
# Output lands under gpurun_out/r02prof/; profiles/make_r02.py turns it into the files kept in profiles/.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r02prof
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_k -o k -- python3 bench.py --steps 10 --warmup 10 --cpu-sample 0 --secondary none > $O/bench_k.json 2> $O/k.err
echo k done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_f -o f -- python3 bench.py --steps 5 --warmup 1 --cpu-sample 0 --secondary none > $O/bench_f.json 2> $O/f.err
echo f done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_w -o w -- python3 bench.py --steps 5 --warmup 1 --cpu-sample 0 --secondary none > $O/bench_w.json 2> $O/w.err
echo w done
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/prof_m -o m -- python3 bench.py --steps 5 --warmup 1 --cpu-sample 0 --secondary none > $O/bench_m.json 2> $O/m.err
echo m done
python3 profiles/summarize.py $O $O/summary > /dev/null
# per-launch durations of the roofline kernel (the full per-dispatch CSV is large)
grep -h "k_gemm_nt_f32_streamk" $O/prof_k/*kernel_trace.csv > $O/gemm_launches.csv || true
head -1 $O/prof_k/*kernel_trace.csv > $O/kernel_trace_header.csv
echo summarized
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fsvi32 -o k -- python3 examples/olfactory_fsvi.py --expansions 300 --growth 100 --dtype f32 > $O/fsvi32.log 2> $O/fsvi32.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fsvi64 -o k -- python3 examples/olfactory_fsvi.py --expansions 300 --growth 100 --dtype f64 > $O/fsvi64.log 2> $O/fsvi64.err
find $O -name '*kernel_trace.csv' -delete
find $O -name '*agent_info.csv' -delete
du -sh $O
