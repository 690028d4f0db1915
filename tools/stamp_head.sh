#!/bin/bash
# Writes build/git_head.txt (HEAD, "+dirty" when tracked files differ from it).  Run HERE before every gpurun call: the snapshot that
# travels to the GPU box has no .git, and bench.py / profiles/summarize.py record the HEAD a measurement was taken at from this file.
cd "$(dirname "$0")/.." || exit 1
mkdir -p build
h=$(git rev-parse HEAD | cut -c1-12)
[ -n "$(git status --porcelain --untracked-files=no)" ] && h="$h+dirty"
echo "$h" > build/git_head.txt
echo "stamped $h"
