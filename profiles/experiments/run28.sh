#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
JAF_WGRAD_NO_DB=1 bash scratch/ab.sh x28_nodb$i | grep -E "ms/step"
bash scratch/ab.sh x28_db$i | grep -E "ms/step"
done
