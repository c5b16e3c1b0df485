#!/bin/bash
# Builds variants/lib_<name>.so from a git revision's csrc/ (or the working tree with rev = WORK): A/B partner for scripts/ab_r04.sh.
# Usage: bash scripts/build_variant.sh <name> <rev|WORK> [extra hipcc flags]
name=$1; rev=$2; shift 2
dir=/tmp/variant_$name; rm -rf $dir; mkdir -p $dir variants
if [ "$rev" == "WORK" ]; then cp doppel-speller_amd/csrc/*.hip doppel-speller_amd/csrc/*.h doppel-speller_amd/csrc/*.inc $dir/; else
  for f in $(git ls-tree --name-only $rev doppel-speller_amd/csrc/); do git show $rev:$f > $dir/$(basename $f); done; fi
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -I include "$@" -o variants/lib_$name.so $dir/*.hip && echo built variants/lib_$name.so
