# Sourced by every tools/gpu/r04_*.sh (run on the GPU box through gpurun).
#  * every run writes into its OWN directory, gpurun_out/<script>_<UTC time>_<pid>/ -- a retry never overwrites the logs of the run before it;
#  * keep_failure <log...>: copies the logs of a step that failed into gpurun_out/failures/<run>/ at once, before anything else runs
#    (gpurun merges gpurun_out/ back; the failing run's records are then copied under profiles/ by hand and committed).
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
RUN_NAME="$(basename "${BASH_SOURCE[1]:-run}" .sh)_$(date -u +%Y%m%dT%H%M%S)_$$"
O="gpurun_out/$RUN_NAME"
mkdir -p "$O"
echo "run directory: $O"
keep_failure() {
  mkdir -p "gpurun_out/failures/$RUN_NAME"
  for f in "$@"; do [ -e "$f" ] && cp -r "$f" "gpurun_out/failures/$RUN_NAME/"; done
  (rocm-smi --showuse --showmemuse 2>/dev/null | head -20; dmesg 2>/dev/null | tail -20) > "gpurun_out/failures/$RUN_NAME/box_state.txt" 2>&1
  echo "FAILED STEP: logs kept in gpurun_out/failures/$RUN_NAME/"
}
build_all() {
  python3 -c "import __graft_entry__ as g; g.build()" > "$O/build.log" 2>&1 || { tail -20 "$O/build.log"; keep_failure "$O/build.log"; exit 1; }
}
