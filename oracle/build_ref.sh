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
# flang's warnings about the reference sources are noise; a failing compile is repeated with its messages shown
fc() { "$FC" "$@" 2>/dev/null || "$FC" "$@"; }
for f in mod_atparam mod_spectral mod_fft spe_spectral spe_subfft_fftpack; do
  fc $FLAGS -I"$REF" -c "$REF/$f.f90" -o "$f.o"
done
fc $FLAGS -c "$HERE/ref_spectral_driver.f90" -o ref_spectral_driver.o
"$FC" -shared -o "$OUT/libref_spectral.so" ref_spectral_driver.o spe_spectral.o spe_subfft_fftpack.o mod_atparam.o mod_spectral.o mod_fft.o
echo "build_ref: wrote $OUT/libref_spectral.so"

# ---- dynamical-core subset (tables + spectral-space routines): pure Fortran, no dependency outside these files ----
# dyn_step.f90 defines hordif/timint but also step(), whose calls to grtend (-> phypar, the column physics: out of scope,
# not built) stay unresolved.  They are never called; the two references are made weak so the library loads.
for f in mod_tsteps mod_dyncon0 mod_dyncon1 mod_dyncon2 mod_hdifcon mod_dynvar spe_matinv dyn_geop dyn_sptend dyn_implic dyn_step ini_indyns ini_impint; do
  fc $FLAGS -I"$REF" -c "$REF/$f.f90" -o "$f.o"
done
OBJCOPY=/opt/rocm/lib/llvm/bin/llvm-objcopy
"$OBJCOPY" --weaken-symbol=grtend_ dyn_step.o
fc $FLAGS -c "$HERE/ref_dyn_driver.f90" -o ref_dyn_driver.o
"$FC" -shared -o "$OUT/libref_dyn.so" ref_dyn_driver.o dyn_geop.o dyn_sptend.o dyn_implic.o dyn_step.o ini_indyns.o ini_impint.o spe_matinv.o \
    spe_spectral.o spe_subfft_fftpack.o mod_atparam.o mod_spectral.o mod_fft.o mod_tsteps.o mod_dyncon0.o mod_dyncon1.o mod_dyncon2.o mod_hdifcon.o mod_dynvar.o
echo "build_ref: wrote $OUT/libref_dyn.so"

# ---- column physics: the reference's parametrisation routines depend only on mod_atparam, mod_physcon and their
# own constant modules -- compiled in place, no stand-ins.  phypar itself (driver of the grid transforms, coupler fluxes, SPPT)
# is NOT built: the harness exposes the individual routines it calls.
# ini_fordate.f90 (the per-window forcing set-up: albedos, tcorh, qcorh) reads plain data modules only and calls spec / shtorh /
# radset / sflset / sol_oz, all of which are compiled here already -- compiled in place as well.
for f in mod_physcon mod_cnvcon mod_lsccon mod_radcon mod_sflcon mod_vdicon phy_convmf phy_lscond phy_shtorh phy_radiat phy_suflux phy_vdifsc ini_inphys \
         mod_lflags mod_surfcon mod_cli_land mod_cli_sea mod_var_land mod_var_sea mod_date ini_fordate; do
  fc $FLAGS -I"$REF" -c "$REF/$f.f90" -o "$f.o"
done
fc $FLAGS -c "$HERE/ref_phy_driver.f90" -o ref_phy_driver.o
"$FC" -shared -o "$OUT/libref_phy.so" ref_phy_driver.o phy_convmf.o phy_lscond.o phy_shtorh.o phy_radiat.o phy_suflux.o phy_vdifsc.o ini_inphys.o \
    ini_fordate.o mod_lflags.o mod_surfcon.o mod_cli_land.o mod_cli_sea.o mod_var_land.o mod_var_sea.o mod_date.o mod_tsteps.o mod_dyncon0.o mod_hdifcon.o \
    spe_spectral.o spe_subfft_fftpack.o mod_spectral.o mod_fft.o \
    mod_atparam.o mod_physcon.o mod_cnvcon.o mod_lsccon.o mod_radcon.o mod_sflcon.o mod_vdicon.o
echo "build_ref: wrote $OUT/libref_phy.so"
