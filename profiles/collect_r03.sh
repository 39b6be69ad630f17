# Round-3 evidence, collected on the MI355X box.  One rocprofv3 pass per counter set (gpurun refuses --pmc combined with
# other traces), the program directly behind `--`, environment exported beforehand:
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash profiles/collect_r03.sh'
# Output: gpurun_out/r03prof/<config>/{bench_k.json, k_kernel_stats.csv, launches.csv, pmc.json}; profiles/make_r03.py turns it
# into the files kept under profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r03prof
rm -rf $O && mkdir -p $O
prof() {   # tag, kernel pattern of the dominant kernel, bench args...
  tag=$1; pat=$2; shift 2
  D=$O/$tag
  mkdir -p $D
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $D/k -o k -- python3 bench.py "$@" > $D/bench_k.json 2> $D/k.err
  echo "$tag k done"
  timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/f -o f -- python3 bench.py "$@" > $D/bench_f.json 2> $D/f.err
  echo "$tag f done"
  timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/w -o w -- python3 bench.py "$@" > $D/bench_w.json 2> $D/w.err
  echo "$tag w done"
  timeout -k 10 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $D/m -o m -- python3 bench.py "$@" > $D/bench_m.json 2> $D/m.err
  echo "$tag m done"
  python3 profiles/tools/pmc_pack.py $D "$pat" > $D/pack.log 2>&1
  cp $D/k/*kernel_stats.csv $D/k_kernel_stats.csv
  head -1 $D/k/*kernel_trace.csv > $D/launches.csv
  grep -h "$pat" $D/k/*kernel_trace.csv >> $D/launches.csv || true
  find $D -name '*kernel_trace.csv' -delete; find $D -name '*counter_collection.csv' -delete; find $D -name '*agent_info.csv' -delete
  rm -rf $D/f $D/w $D/m
}
COMMON="--cpu-sample 0 --secondary none"
prof c4 k_gemm_nt_f32_streamk_fused $COMMON --steps 10 --warmup 10
prof c4_r5 k_gemm_nt_f32_streamk $COMMON --reach 5 --blocks 1 --steps 10 --warmup 5
prof c4_f64 k_gemm_nt_f32_streamk_fused $COMMON --dtype f64 --blocks 1 --steps 10 --warmup 5
export PBVI_F64_SCREEN=off
prof c4_f64_pure k_gemm_nt_f64_mfma $COMMON --dtype f64 --blocks 1 --steps 5 --warmup 2
unset PBVI_F64_SCREEN
export PBVI_GEMM_DENSE=1
prof c3_dense k_gemm_nt_f32_mfma $COMMON --mode dense --blocks 1 --steps 3 --warmup 1
unset PBVI_GEMM_DENSE
# the solve protocol end to end (SolverHistory's own timings), and the one-rank RCCL rehearsal of the sharded step
python3 examples/olfactory_fsvi.py --expansions 300 --growth 100 --dtype f32 > $O/fsvi300_f32.log 2> $O/fsvi300_f32.err
python3 examples/olfactory_fsvi.py --expansions 300 --growth 100 --dtype f64 > $O/fsvi300_f64.log 2> $O/fsvi300_f64.err
echo fsvi done
PBVI_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python3 bench.py --cpu-sample 0 --secondary none --steps 20 > $O/dist_1rank_rccl.json 2> $O/dist_1rank.err
echo dist done
python3 bench.py --steps 50 --warmup 10 > $O/bench_default.json 2> $O/bench_default.err
echo bench done
du -sh $O
