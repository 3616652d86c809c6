#!/bin/bash
SRHIP_LIB=$(pwd)/sr-caco-2_amd/lib/libsrhip_exp.so timeout 300 python tools/mb_nt_stamps.py 2>&1 | tail -12
