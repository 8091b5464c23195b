for kb in 1536 3072; do for bpc in 2 3 4; do for ch in 1024 2048 4096; do
  echo -n "slice_kb=$kb bpc=$bpc chunk=$ch: "; NQE_JOIN_PART_SLICE_KB=$kb NQE_JOIN_PART_PLACE_BPC=$bpc NQE_JOIN_PART_CHUNK=$ch python tools/probe_build.py 100000000 2>&1 | grep "random    keys, key + int payload  \[partitioned\]" | sed 's/.*partitioned\]//'
done; done; done
