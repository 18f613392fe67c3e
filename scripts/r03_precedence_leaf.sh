# critical-path precedence leaf (one MI355X): the leaf alone and inside the list policy, LDS scratch (default at these sizes)
for cfg in "10 5 2048 20 2" "20 10 2048 10 2" "50 20 1024 3 2"; do
  echo "leaf alone: $(timeout 600 python scripts/precedence_bench.py $cfg precedence 2>&1 | tail -1 | cut -c1-700)"
  echo "policy:     $(timeout 600 python scripts/precedence_bench.py $cfg policy 2>&1 | tail -1 | cut -c1-700)"
  echo "change+swap:$(timeout 600 python scripts/precedence_bench.py $cfg 2>&1 | tail -1 | cut -c1-700)"
done
