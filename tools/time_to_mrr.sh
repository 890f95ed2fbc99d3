#!/bin/bash
# tools/time_to_mrr.sh [extra dglke_train flags]: FB15k-shaped planted graph -> TransE_l2 with the reference's
# FB15k hyper-parameters (examples/fb15k/multi_gpu.sh: batch 1000, neg 200, dim 400, gamma 19.9, lr 0.25, -adv,
# rc 1e-9), validation every 500 steps, stop at MRR >= 0.65.  Prints the time-to-MRR line.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/.." && pwd)
D=/tmp/fb15k_planted
[ -f $D/train.txt ] || python $R/tools/make_planted_fb15k.py $D $GEN_ARGS
python $R/dgl-ke_amd/dglke_train --model_name TransE_l2 --format udd_hrt --dataset fb15k_planted --data_path $D \
  --data_files entities.dict relations.dict train.txt valid.txt test.txt --save_path /tmp/ckpts --no_save_emb --gpu 0 \
  --batch_size 1000 --neg_sample_size 200 --hidden_dim 400 --gamma 19.9 --lr 0.25 -adv --regularization_coef 1e-9 \
  --max_step 24000 --log_interval 1000 --eval_interval 500 --valid --test --target_mrr 0.65 --graph_steps 100 "$@"
