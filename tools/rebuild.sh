#!/bin/bash
# Rebuild libmzmcts.so (stale objects only) and, with "stamps", the -DMZ_STAMPS diagnostic library of tools/stamp_fused.py.
set -e
cd "$(dirname "$0")/.."
python -c "
import importlib, sys
sys.path.insert(0, '.')
print(importlib.import_module('muzero-hypermodel_amd.build').build_native())"
if [ "$1" = "stamps" ]; then MZ_STAMPS_BUILD_ONLY=1 python tools/stamp_fused.py; ls -la tools/_stamps/libmzmcts.so; fi
