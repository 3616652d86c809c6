#!/bin/bash
bash tools/refresh_profiles.sh ${1:-swinir_x8} gpurun_out/prof_${1:-swinir_x8} 2>&1 | tail -40
