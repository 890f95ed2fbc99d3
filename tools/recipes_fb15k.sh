#!/bin/bash
# the reference's single-GPU FB15k recipes (examples/fb15k/multi_gpu.sh) through dglke_train on the FB15k-shaped PLANTED graph: steady-state
# us/step (second log interval), seconds of the final test evaluation, test MRR.  usage: tools/recipes_fb15k.sh [steps]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; S=${1:-3000}
D=/tmp/fb15k_planted; [ -f $D/train.txt ] || python $R/tools/make_planted_fb15k.py $D > /dev/null 2>&1
run() {  # name, flags...
  n=$1; shift
  out=$(python $R/dgl-ke_amd/dglke_train --format udd_hrt --dataset fb15k_planted --data_path $D --data_files entities.dict relations.dict train.txt valid.txt test.txt \
        --save_path /tmp/ckpts --no_save_emb --gpu 0 --log_interval 1000 --batch_size_eval 16 --test -adv --max_step $S "$@" 2>&1)
  us=$(echo "$out" | grep "\[Train\] 1000 steps take" | sed -n 2p | sed -E 's/.*take ([0-9.]+) seconds.*/\1/')
  ts=$(echo "$out" | grep "testing takes" | sed -E 's/.*takes ([0-9.]+) seconds.*/\1/')
  mrr=$(echo "$out" | grep "Test average MRR" | sed -E 's/.*MRR: ([0-9.]+).*/\1/' | cut -c1-6)
  printf "%-34s %8s ms per 1000 steps   test %6s s   MRR %s\n" "$n" "$(python -c "print(round(1000*float('${us:-nan}'),1))")" "${ts:-?}" "${mrr:-?}"
  [ -z "$us" ] && echo "$out" | tail -5
}
run "TransE_l1  b1000 n200 d400"        --model_name TransE_l1 --batch_size 1000 --neg_sample_size 200 --regularization_coef 1e-07 --hidden_dim 400 --gamma 16.0 --lr 0.01
run "TransE_l2  b1000 n200 d400"        --model_name TransE_l2 --batch_size 1000 --neg_sample_size 200 --regularization_coef=1e-9 --hidden_dim 400 --gamma 19.9 --lr 0.25
run "DistMult   b1000 n200 d400"        --model_name DistMult --batch_size 1000 --neg_sample_size 200 --hidden_dim 400 --gamma 143.0 --lr 0.08
run "ComplEx    b1000 n200 d400"        --model_name ComplEx --batch_size 1000 --neg_sample_size 200 --hidden_dim 400 --gamma 143.0 --lr 0.1 --regularization_coef 2.00E-06
run "RESCAL     b1000 n200 d500"        --model_name RESCAL --batch_size 1000 --neg_sample_size 200 --hidden_dim 500 --gamma 24.0 --lr 0.03
run "RotatE     b2048 n256 d200 -de nds" --model_name RotatE --batch_size 2048 --neg_sample_size 256 --regularization_coef 1e-07 --hidden_dim 200 --gamma 12.0 --lr 0.009 -de --neg_deg_sample
run "TransR     b1000 n200 d200"        --model_name TransR --batch_size 1000 --neg_sample_size 200 --regularization_coef 5e-8 --hidden_dim 200 --gamma 8.0 --lr 0.015
