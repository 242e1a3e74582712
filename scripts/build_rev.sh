#!/bin/sh
# Builds the library of another revision as a development variant (for same-call A/B runs against the working tree):
#   scripts/build_rev.sh <git-rev> <name> [extra hipcc flags]   -> nano-kazen_amd/csrc/variants/<name>/libkazen_mi355x.so
set -e
REV=$1; NAME=$2; shift 2
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
WT=$(mktemp -d /tmp/kzrev.XXXXXX)
git -C "$ROOT" worktree add --detach "$WT" "$REV" > /dev/null
sh "$WT/scripts/build_variant.sh" "$NAME" "$@"
mkdir -p "$ROOT/nano-kazen_amd/csrc/variants/$NAME"
cp "$WT/nano-kazen_amd/csrc/variants/$NAME/libkazen_mi355x.so" "$ROOT/nano-kazen_amd/csrc/variants/$NAME/"
git -C "$ROOT" worktree remove --force "$WT"
echo "built variants/$NAME from $REV"
