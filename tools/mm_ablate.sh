#!/bin/bash
# development: k_schur_mm stamps under ablation macros (wrong results by design), one build per entry of $ABLS
set -e
for abl in ${ABLS:-NONE SRK_MM_ABL_NOLDS SRK_MM_ABL_NOHELP SRK_MM_ABL_NOLDS+SRK_MM_ABL_NOHELP}; do
  echo "=== $abl"
  EXTRA="-D${abl//+/ -D}" bash "$GRAFT_REPO_ROOT/tools/mm_stamps.sh"
done
