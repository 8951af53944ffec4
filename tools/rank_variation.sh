#!/bin/bash
# which launch configuration varies from process to process?  md5 of every shape's predicted cloud, 3 runs each.
R=$PWD; export PYTHONPATH=$R BDM_DIST_BACKEND=gloo BDM_SHARE_GPU=1
ARGS="run.job=sample_bdm_blending run.rng=per_shape dataset=synthetic dataset.max_points=1024 dataset.num_shapes=4 dataloader.batch_size=2 aux_run.roll_step=1 aux_run.milestones=[1000,998,996,995] run.name=v"
for w in 1 2 1 2 1 2 2 2; do
  d=/tmp/var_$RANDOM; rm -rf $d
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port $((29800 + RANDOM % 100)) main_blending.py $ARGS run.save_dir=$d $EXTRA > /dev/null 2>&1
  echo "world=$w $(cd $d/v/*/sample_bdm_blending/pred/chair && md5sum *.ply | cut -c1-8 | tr '\n' ' ')"
done
