#!/bin/bash
# tools/time_to_mrr.sh [extra dglke_train flags]: time-to-MRR@0.65 of TransE_l2 with the reference's FB15k hyper-parameters
# (examples/fb15k/multi_gpu.sh: batch 1000, neg 200, dim 400, gamma 19.9, lr 0.25, -adv, rc 1e-9), validation every 500
# steps, stop at MRR >= 0.65.
#   * if an operator-supplied REAL FB15k directory exists (FB15K_DIR, default data/FB15k next to the repo or under
#     $KGE_DATA_PATH: entities.dict relations.dict train.txt valid.txt test.txt, the reference's built-in layout) it is used
#     and the line says so - this is BASELINE.json's metric;
#   * otherwise an FB15k-shaped PLANTED graph is generated (no network here): the run is labelled as such and only the step
#     and evaluation rates carry over.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(cd "$(dirname "$0")/.." && pwd)
REAL=${FB15K_DIR:-${KGE_DATA_PATH:-$R/data}/FB15k}
COMMON="--model_name TransE_l2 --save_path /tmp/ckpts --no_save_emb --gpu 0 --batch_size 1000 --neg_sample_size 200 \
  --hidden_dim 400 --gamma 19.9 --lr 0.25 -adv --regularization_coef 1e-9 --max_step 24000 --log_interval 1000 \
  --eval_interval 500 --valid --test --target_mrr 0.65 --graph_steps 100 --batch_size_eval 16"
if [ -f "$REAL/train.txt" ] && [ -f "$REAL/entities.dict" ]; then
  echo "# time_to_mrr: REAL FB15k files found in $REAL (operator-supplied) - this run measures BASELINE.json's metric"
  python $R/dgl-ke_amd/dglke_train --dataset FB15k --data_path "$(dirname "$REAL")" $COMMON "$@"
else
  echo "# time_to_mrr: no FB15k files under $REAL - PLANTED FB15k-shaped graph (not the BASELINE metric; rates only)"
  D=/tmp/fb15k_planted
  [ -f $D/train.txt ] || python $R/tools/make_planted_fb15k.py $D $GEN_ARGS
  python $R/dgl-ke_amd/dglke_train --format udd_hrt --dataset fb15k_planted --data_path $D \
    --data_files entities.dict relations.dict train.txt valid.txt test.txt $COMMON "$@"
fi
