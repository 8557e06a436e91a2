set -e
OUT=gpurun_out/profiles_new
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# the kernel-trace half of tools/make_profiles.sh (the trace must not contain the teacher-cache leg)
sed -n '/^rm -rf gpurun_out\/prof_kt$/,/^python3 tools\/step_timeline.py/p' tools/make_profiles.sh > /tmp/kt_part.sh
bash /tmp/kt_part.sh
