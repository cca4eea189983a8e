#!/bin/bash
# tools/mkab.sh <name> [git-rev]   build the native libraries of a revision (default: working tree) into ab/<name>/
set -e
name=$1; rev=$2
cd /root/repo
mkdir -p ab/$name
if [ -n "$rev" ]; then
  rm -rf /tmp/abtree && mkdir -p /tmp/abtree && git archive $rev | tar -x -C /tmp/abtree
  (cd /tmp/abtree && python -c "
import sys; sys.path.insert(0,'.')
import importlib
b=importlib.import_module('k-wave-fluid-cuda_amd.build'); b.build_hip(); b.build_host()
try: b.build_host_h5()
except Exception as e: print('h5 skipped', e)
")
  cp /tmp/abtree/k-wave-fluid-cuda_amd/lib/*.so ab/$name/
else
  python /tmp/buildit.py > /dev/null
  cp k-wave-fluid-cuda_amd/lib/*.so ab/$name/
fi
ls ab/$name
