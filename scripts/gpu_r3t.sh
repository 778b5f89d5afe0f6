#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
for sz in tiny mid; do
TAG="$sz normal " timeout 200 python scripts/dev/ppo_determinism.py $sz 2>&1 | grep -v amdgpu.ids | cut -c1-220
TAG="$sz serial " SERIAL=1 timeout 200 python scripts/dev/ppo_determinism.py $sz 2>&1 | grep -v amdgpu.ids | cut -c1-220
TAG="$sz fresh  " IPLAN_FRESH_BATCH=1 timeout 200 python scripts/dev/ppo_determinism.py $sz 2>&1 | grep -v amdgpu.ids | cut -c1-220
TAG="$sz nodefer" NO_DEFER=1 timeout 200 python scripts/dev/ppo_determinism.py $sz 2>&1 | grep -v amdgpu.ids | cut -c1-220
TAG="$sz ser+nd " SERIAL=1 NO_DEFER=1 timeout 200 python scripts/dev/ppo_determinism.py $sz 2>&1 | grep -v amdgpu.ids | cut -c1-220
done
