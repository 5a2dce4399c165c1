for off in small sigma1 sigma1.4 sigma2; do for mg in 2 3 4; do
  echo "== offsets=$off margin=$mg"; timeout 120 python profiles/dcn_layer.py --offsets $off --margin $mg --iters 3 --time 2>&1 | grep -E "B=|dcn_bwd_data|dcnw_fwd"
done; done
