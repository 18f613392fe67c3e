for s in "10 5 4096" "15 5 4096" "20 5 4096" "12 10 2048" "20 10 2048" "30 10 2048" "40 10 2048" "15 15 2048" "20 20 2048"; do set -- $s
  SF_AMD_DEBUG_LAUNCH=1 timeout 200 python scripts/precedence_solve60.py 4 $1 $2 $3 feasible policy 2>&1 | python -c "
import sys,json,re
T='?'
for l in sys.stdin:
    if l.startswith('[sf]'):
        m=re.search(r'prec groups (\d+)', l); T=m.group(1) if m else T
    elif l.startswith('{'):
        d=json.loads(l); print('$1x$2 R$3 default groups', T, '%.1f M moves/s' % (d['gpu']['moves_per_s']/1e6), d['gpu']['best_score'])"
done
