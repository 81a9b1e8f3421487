#!/usr/bin/env bash
# Cholesky single solve: CUs the trailing stream leaves free (SML_LU_RESERVE) with an UNCONFINED panel stream (SML_LU_CONFINE=0).
set -e
for r in 0 16 32 48 64 96 128; do
  echo "reserve $r"; SML_LU_CONFINE=0 SML_LU_RESERVE=$r timeout -k 10 120 python profiles/micro/fit_solvers.py chol 5
done
