#!/bin/bash
# Same-box A/B of two builds of the library (build/ab/prev.so against build/ab/new.so; both travel with the snapshot): the file the
# package loads is swapped between runs of the same command.  usage: tools/ab_two_builds.sh <rounds> <command...>
rounds=$1; shift
cp gnnome_amd/lib/libgnnome_hip.so /tmp/shipped.so
for rnd in $(seq 1 $rounds); do
  for b in prev new; do
    cp build/ab/$b.so gnnome_amd/lib/libgnnome_hip.so
    echo "== round $rnd build $b"
    "$@"
  done
done
cp /tmp/shipped.so gnnome_amd/lib/libgnnome_hip.so
