# Same command line as the reference's Classification/cmd/run_perturb.sh (ResNet-56s, K=5, idx 13, gamma 0.5).
# Run from cv_a-fan_amd/ like the reference runs from Classification/.  Data parallel on one 8xMI355X node:
#   python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 main_perturb.py <same flags>
python -u main_perturb.py --seed 3 --save_dir res56s_perturb_pgd5_layer1_gamma0.5_norandclip --gamma 0.5
