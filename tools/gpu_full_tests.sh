#!/bin/bash
# On a GPU box (via gpurun, from the repo root): the full `-m gpu` suite as the driver runs it, recording every gradient
# tensor's acceptance route (-> gpurun_out/routes_new.json; review, then copy to tests/golden/full/routes.json), then smoke().
mkdir -p gpurun_out
rm -f gpurun_out/routes_new.json
SHINEON_WRITE_ROUTES=$PWD/gpurun_out/routes_new.json SHINEON_ROUTES_NOCHECK=${ROUTES_NOCHECK-1} \
  python -m pytest tests -q -m gpu --durations=15 -o faulthandler_timeout=240 -p no:cacheprovider > gpurun_out/${TAG:-r05}_gpu_tests.log 2>&1
tail -40 gpurun_out/${TAG:-r05}_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
