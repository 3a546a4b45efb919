#!/bin/bash
# Two / three worker contexts on whole strict --fs passes with the regions' Forward packed onto fewer CUs (BATH_HIP_FS_REGION_CU_SHARE)
cd $GRAFT_REPO_ROOT
for s in 1 2 4 1 2 4; do
  echo "== BATH_HIP_FS_REGION_CU_SHARE=$s"
  BATH_HIP_FS_REGION_CU_SHARE=$s python3 tools/fs_workers_probe.py --workers 1,2 --passes 6 2>&1 | grep "^workers"
done
