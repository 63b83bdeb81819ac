#!/bin/bash
# dev: builds variant libraries for tools/dev_ab.sh in parallel:  tools/dev_build_variants.sh name1:"-DX=1 -DY=2" name2:"" ...
# -> 4dgs-slam_amd/_variants/<name>.so (git-ignored; they travel to the GPU box with the snapshot)
mkdir -p /root/repo/4dgs-slam_amd/_variants
cd /root/repo/4dgs-slam_amd/csrc
pids=()
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( ./build.sh $flags -o ../_variants/$name.so > /tmp/variant_$name.log 2>&1 || echo "FAILED $name (see /tmp/variant_$name.log)" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
ls -la ../_variants/
