cd nerf-hugs_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-inline-asm -munsafe-fp-atomics"
objs=$(ls _obj/*.o | grep -v fieldfuse)
for v in "$@"; do
  defs=$(echo $v | tr '+' '\n' | sed 's/^/-DFF_/' | tr '\n' ' ')
  /opt/rocm/bin/hipcc $F $defs -c hugs_fieldfuse.hip -o /tmp/ff_$v.o &
done
wait
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/libhugs_$v.so $objs /tmp/ff_$v.o
done
