# Copy the records of tools/final_round.sh from gpurun_out/final/ into profiles/ and condense the rocprofv3 output.
set -u
R=${1:-r03}
for f in gpurun_out/final/${R}_*; do [ -s "$f" ] && cp "$f" profiles/; done
python tools/summarize_prof.py ${R}_train_bf16
python tools/summarize_prof.py ${R}_forward
ls profiles | grep "^${R}_"
