#!/bin/bash
# Rebuild libdoppel_amd.so (and the oracle); exits non-zero when the compile fails, prints only errors.
cd "$(dirname "$0")/.." || exit 1
python -c "import __graft_entry__ as g; g.build()" > /tmp/ds_build.log 2>&1
status=$?
if [ $status -ne 0 ]; then grep -B2 -A8 "error" /tmp/ds_build.log | head -60; echo "BUILD FAILED"; exit $status; fi
echo "build ok"
