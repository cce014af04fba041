#!/bin/bash
# Build libscoary_hip.so from the kernel sources of ANOTHER git revision (or of the working tree with extra
# -D flags) into _ab/<name>.so, for same-box A/B runs (SCOARY_HIP_LIB=_ab/<name>.so selects it; _ab/ is
# git-ignored but travels to the GPU box).   tools/build_alt.sh <name> <git-rev|WORKTREE> [extra hipcc flags]
set -e
cd "$(dirname "$0")/.."
NAME=$1; REV=$2; shift 2
mkdir -p _ab/src_$NAME/csrc _ab/src_$NAME/include
if [ "$REV" = WORKTREE ]; then
  cp scoary_amd/csrc/*.hip scoary_amd/csrc/*.hpp scoary_amd/csrc/*.inc _ab/src_$NAME/csrc/
  cp include/*.h _ab/src_$NAME/include/
else
  for f in $(git ls-tree --name-only $REV scoary_amd/csrc/ | grep -E '\.(hip|hpp|inc)$'); do git show $REV:$f > _ab/src_$NAME/csrc/$(basename $f); done
  for f in $(git ls-tree --name-only $REV include/); do git show $REV:$f > _ab/src_$NAME/include/$(basename $f); done
fi
S=_ab/src_$NAME/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -I_ab/src_$NAME/include "$@" \
  $S/scoary_context.hip $S/scoary_assoc.hip $S/scoary_lists.hip $S/scoary_listbuild.hip $S/scoary_labels.hip $S/scoary_tree.hip \
  -o _ab/$NAME.so
ls -la _ab/$NAME.so
