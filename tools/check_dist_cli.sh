set -e
D=$(mktemp -d)
python - <<PY
import gzip, os
for fn in ("Gene_presence_absence.csv","Tetracycline_resistance.csv"):
    open(os.path.join("$D", fn),"w").write(gzip.open("tests/golden/exampledata/%s.gz"%fn,"rt").read())
PY
SCOARY_EXERCISE_DIST=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 -m scoary_amd -g $D/Gene_presence_absence.csv -t $D/Tetracycline_resistance.csv --no_pairwise -e 100 -o $D/out --no-time 2>&1 | tail -5
python -m scoary_amd -g $D/Gene_presence_absence.csv -t $D/Tetracycline_resistance.csv --no_pairwise -e 100 -o $D/out1 --no-time > /dev/null 2>&1
cmp $D/out/Tetracycline_resistance.results.csv $D/out1/Tetracycline_resistance.results.csv && echo SAME_OUTPUT
