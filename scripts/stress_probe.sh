#!/bin/bash
# stress_probe.sh [SEED] - random shapes through the C probe with the planner free to pick tier and orientation; every
# cell checked against the CPU oracle.  Torch-free: ~1 s per shape on the GPU box.
P=tests/native/bin/systolic_probe
RANDOM=${1:-12345}
export PROBE_ALARM=60
unset SZS_ROCM_TIER SZS_ROCM_SWAP
fails=0
for i in $(seq 1 ${2:-48}); do
    family=$(echo lev levw nw sw | cut -d' ' -f$((RANDOM % 4 + 1)))
    case $((RANDOM % 6)) in
        0) q=$((RANDOM % 4 + 1)); c=$((RANDOM % 4 + 1)); lo=$((RANDOM % 3000)); hi=$((lo + RANDOM % 6000));;
        1) q=$((RANDOM % 40 + 1)); c=$((RANDOM % 40 + 1)); lo=$((RANDOM % 300)); hi=$((lo + RANDOM % 1500));;
        2) q=$((RANDOM % 300 + 1)); c=$((RANDOM % 6 + 1)); lo=0; hi=$((RANDOM % 400 + 1));;
        3) q=$((RANDOM % 6 + 1)); c=$((RANDOM % 600 + 1)); lo=0; hi=$((RANDOM % 300 + 1));;
        4) q=$((RANDOM % 200 + 1)); c=$((RANDOM % 200 + 1)); lo=$((RANDOM % 100)); hi=$((lo + RANDOM % 200));;
        5) q=1; c=1; lo=$((RANDOM % 20000)); hi=$((lo + RANDOM % 20000));;
    esac
    if [ "$family" = sw ] || [ "$family" = nw ]; then gaps="$((-(RANDOM % 9) - 1)) $((-(RANDOM % 4) - 1))"; else gaps=""; fi
    out=$(timeout 120 $P $family $q $c $lo $hi 1 $gaps 2>&1 | tail -1)
    echo "$out" | cut -c1-200
    echo "$out" | grep -q " 0 bad cells" || fails=$((fails + 1))
done
echo "stress: $fails failing shapes"
