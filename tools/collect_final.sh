# Copy the records of tools/final_round.sh from gpurun_out/final/ into profiles/ and condense the rocprofv3 output.
set -u
R=${1:-r03}
for f in gpurun_out/final/${R}_*; do [ -s "$f" ] && cp "$f" profiles/; done
python tools/summarize_prof.py ${R}_train_bf16
python tools/summarize_prof.py ${R}_forward
for t in forward_bf16 longform longform_bf16 train_f16x3; do [ -d gpurun_out/prof_${R}_$t ] && python tools/summarize_prof.py ${R}_$t; done
ls profiles | grep "^${R}_"
