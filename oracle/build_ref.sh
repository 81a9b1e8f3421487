#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY.  Builds oracle/_ref/libref_spectral.so from the reference's own
# spectral sources, compiled IN PLACE under /root/reference/src (nothing is copied into the repo),
# plus the forwarding harness oracle/ref_spectral_driver.f90.  No stand-in modules are written:
# these five reference files have no dependency outside themselves.
#   flags: -fdefault-real-8 == the reference's own promotion flag (src/makefile:6,12 -r8 / -fdefault-real-8)
# Skips silently (exit 0) when /root/reference or amdflang is absent (e.g. on the GPU box, which
# only ever uses the prebuilt .so that travels with the snapshot).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${SML_REFERENCE_SRC:-/root/reference/src}"
FC="${SML_FC:-/opt/rocm/bin/amdflang}"
OUT="$HERE/_ref"
if [ ! -d "$REF" ] || [ ! -x "$FC" ]; then
  echo "build_ref: reference sources or amdflang not present; keeping prebuilt oracle/_ref (if any)"; exit 0
fi
mkdir -p "$OUT/obj"
cd "$OUT/obj"
FLAGS="-fdefault-real-8 -O2 -fPIC"
for f in mod_atparam mod_spectral mod_fft spe_spectral spe_subfft_fftpack; do
  "$FC" $FLAGS -I"$REF" -c "$REF/$f.f90" -o "$f.o" 2>/dev/null
done
"$FC" $FLAGS -c "$HERE/ref_spectral_driver.f90" -o ref_spectral_driver.o 2>/dev/null
"$FC" -shared -o "$OUT/libref_spectral.so" ref_spectral_driver.o spe_spectral.o spe_subfft_fftpack.o mod_atparam.o mod_spectral.o mod_fft.o
echo "build_ref: wrote $OUT/libref_spectral.so"
