#!/bin/bash
# Same-box A/B of two builds of the library: tools/ab_lib.sh OTHER.so -- <command ...>
# runs the command alternately with VITK_LIB=OTHER.so and with the in-tree library, twice each.
OTHER=$1; shift; shift
for i in 1 2; do
    echo "== other ($OTHER)"; VITK_LIB=$OTHER timeout -k 10 200 "$@" 2>&1 | grep -v amdgpu.ids || exit 1
    echo "== in-tree";        timeout -k 10 200 "$@" 2>&1 | grep -v amdgpu.ids || exit 1
done
