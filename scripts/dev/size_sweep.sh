cd $GRAFT_REPO_ROOT
for cfg in "512 512 426667" "640 480 500000" "768 512 640000" "1024 512 853333"; do
  set -- $cfg
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --width $1 --height $2 --gaussians $3 2>/dev/null | tail -1 > /tmp/o.json
  python - "$1x$2" <<'PY'
import sys, json
d = json.load(open('/tmp/o.json')); s = d["roofline"]["stage_ms"]; R = d["config"]["num_rendered"]
print(sys.argv[1], "R", R, "fwd us/Minst %.1f" % (s["blend_fwd"] * 1e3 / (R / 1e6)), "bwd us/Minst %.1f" % (s["blend_bwd"] * 1e3 / (R / 1e6)), "ms/step", d["ms_per_step"])
PY
done
