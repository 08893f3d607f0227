#!/bin/bash
mkdir -p gpurun_out/r3c34
timeout 600 python tools/diag_trainer.py > gpurun_out/r3c34/diag.txt 2>&1; grep "sink=" gpurun_out/r3c34/diag.txt | cut -c1-400; tail -3 gpurun_out/r3c34/diag.txt | cut -c1-300
