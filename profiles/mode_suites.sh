# The GPU suite in every advertised mode of the engine: one pytest process per mode, summary
# lines + the names of anything that failed or was skipped into gpurun_out/modes/summary.md.
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash profiles/mode_suites.sh'
cd $GRAFT_REPO_ROOT
O=gpurun_out/modes
rm -rf $O && mkdir -p $O
echo "# GPU suite per engine mode (build $(cat $O/../build_id 2>/dev/null))" > $O/summary.md
echo "" >> $O/summary.md
echo "| mode (environment) | result | seconds |" >> $O/summary.md
echo "|---|---|---|" >> $O/summary.md
run() {
  name=$1; shift
  t0=$(date +%s)
  env "$@" timeout -k 10 900 python -m pytest tests -q -m gpu -rfs -p no:cacheprovider > $O/$name.log 2>&1
  t1=$(date +%s)
  line=$(grep -E "passed|failed|error" $O/$name.log | tail -1)
  echo "| \`$*\` | $line | $((t1-t0)) |" >> $O/summary.md
  echo "$name: $line"
}
run default PBVI_MODE_TAG=default
run belief PBVI_FORMULATION=belief
run nofuse PBVI_NO_FUSED_PROJECT=1
run screen_always PBVI_F64_SCREEN=always
run screen_off PBVI_F64_SCREEN=off
run poison PBVI_POISON=1
run no_l1 PBVI_NO_L1_SCREEN=1
echo "" >> $O/summary.md
echo "## failed / skipped tests per mode" >> $O/summary.md
for f in default belief nofuse screen_always screen_off poison no_l1; do
  echo "" >> $O/summary.md
  echo "### $f" >> $O/summary.md
  echo '```' >> $O/summary.md
  grep -E "^(FAILED|SKIPPED|ERROR)" $O/$f.log >> $O/summary.md || echo "(none)" >> $O/summary.md
  echo '```' >> $O/summary.md
done
cat $O/summary.md
