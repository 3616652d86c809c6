#!/bin/bash
bash tools/step_mfma_pmc.sh gpurun_out/mfma_swinir swinir_x8 2>&1 | tail -14
bash tools/step_mfma_pmc.sh gpurun_out/mfma_edsr8 edsr_x8 2>&1 | tail -10
