#!/usr/bin/env python3
"""Predict the depth of invasion for a directory of Z stacks -- MI355X drop-in for the reference's scripts/compute_inv_depth.py:
same positional arguments, flags and config key, same output (`invasion_depth_predictions.csv`: Z Slice ID, Invasion
Probability, Invasion Prediction (0=no 1=yes); -2, -3 ... when the name is taken), same exit behaviour.

    python compute_inv_depth.py IN_ROOT OUT_ROOT [--channel N] [--time N] [-c CONFIG]

The n_pred_models classifiers with the lowest fine-tuning validation loss (model_training/best_ensemble/best_model_history_*.csv)
are loaded from `best_finetune_weights_{i}.tmatw` (Keras `.h5` files converted once with tools/convert_keras_h5.py --resnet;
TMAT_SYNTHETIC_WEIGHTS=1 runs random-init weights of the same architecture) and run on the GPU through tmat_inv_depth_predict.
Differences: files are read with Pillow (--time / --channel follow the reference's helper.load_image).
"""
import argparse
import csv
import json
import os
import sys
from glob import glob
from pathlib import Path

PKG = Path(__file__).resolve().parents[1]
if str(PKG) not in sys.path:
    sys.path.insert(0, str(PKG))

import numpy as np  # noqa: E402

MODEL_TRAINING_DIR = PKG / "model_training"
DEFAULT_CONFIG_PATH = str(PKG / "config" / "default_invasion_depth_computation.json")
FAIL = "\033[91m[FAILURE]\033[0m"
OK = "\033[92m[SUCCESS]\033[0m"
WARN = "\033[93m[WARNING]\033[0m"


def parse_inv_depth_args(argv=None):
    """same surface as the reference's script_util.parse_inv_depth_args (script_util.py:380-450)"""
    p = argparse.ArgumentParser()
    p.add_argument("in_root", type=str)
    p.add_argument("out_root", type=str)
    p.add_argument("--channel", type=int, default=None)
    p.add_argument("--time", type=int, default=None)
    p.add_argument("-c", "--config", type=str, default=DEFAULT_CONFIG_PATH)
    args = p.parse_args(argv)
    for k, v in vars(args).items():
        if isinstance(v, str):
            setattr(args, k, v.strip("'\""))
    return args


def get_unique_output_filepath(file):
    file = Path(file)
    name, ext = os.path.splitext(file.name)
    n = 1
    while file.exists():
        n += 1
        file = file.parent / f"{name}-{n}{ext}"
    return file


def load_member_weights(best_ensemble_dir: Path, idx: int, last_layer: str):
    from tmat_amd import inv_depth
    h5 = best_ensemble_dir / f"best_finetune_weights_{idx}.h5"
    blob = h5.with_suffix(".tmatw")
    if blob.is_file():
        return blob.read_bytes()
    if os.environ.get("TMAT_SYNTHETIC_WEIGHTS") == "1":
        print(f"[tmat_amd] {h5.name} not found/convertible: using synthetic weights (TMAT_SYNTHETIC_WEIGHTS=1)", flush=True)
        return inv_depth.pack_resnet(inv_depth.synth_resnet_weights(idx, last_layer))
    raise FileNotFoundError(f"{h5}: Keras HDF5 weights must be converted once with tools/convert_keras_h5.py --resnet (expected {blob}); "
                            "set TMAT_SYNTHETIC_WEIGHTS=1 to run with synthetic weights")


def main(argv=None):
    args = parse_inv_depth_args(argv)
    in_root = Path(args.in_root)
    if not in_root.is_dir():
        print(f"{FAIL} Input directory {in_root} does not exist.", flush=True)
        sys.exit(1)
    if not glob(str(in_root / "*")):
        print(f"{FAIL} Input directory is empty: {in_root}", flush=True)
        sys.exit(1)
    out_root = Path(args.out_root)
    if out_root.is_file():
        print(f"{FAIL} Output path is a file: {out_root}", flush=True)
        sys.exit(1)
    try:
        out_root.mkdir(parents=True, exist_ok=True)
    except PermissionError as e:
        print(f"{FAIL} {e}", flush=True)
        sys.exit(1)
    if glob(str(out_root / "*")):
        print(f"{WARN}Output directory is not empty:{os.linesep}\t{out_root}", flush=True)

    with open(MODEL_TRAINING_DIR / "invasion_depth_best_hp.json", "r") as fp:
        best_hp = json.load(fp)
    with open(MODEL_TRAINING_DIR / "invasion_depth_training_values.json", "r") as fp:
        training_values = json.load(fp)
    cls_thresh = training_values["cls_thresh"]
    resnet_inp_shape = tuple(training_values["resnet_inp_shape"])
    n_models = training_values["n_models"]
    last_resnet_layer = best_hp["last_resnet_layer"]
    if not os.path.isfile(args.config):
        print(f"{FAIL} Config file not found: {args.config}", flush=True)
        sys.exit(1)
    with open(args.config, "r", encoding="utf8") as fp:
        config = json.load(fp)
    n_pred_models = config["n_pred_models"]
    if not n_pred_models <= n_models:
        raise AssertionError(f"Desired number of ensemble members ({n_pred_models}) is greater than number of saved models.")
    if resnet_inp_shape[0] != resnet_inp_shape[1] or resnet_inp_shape[2] != 3:
        print(f"{FAIL} resnet_inp_shape {resnet_inp_shape}: only square 3-channel inputs are supported.", flush=True)
        sys.exit(1)

    from tmat_amd import _lib, distributed, helper, inv_depth, zstacks as zs
    best_ensemble_dir = MODEL_TRAINING_DIR / "best_ensemble"
    order = inv_depth.best_model_indices(best_ensemble_dir, n_models, n_pred_models)
    # one process per GPU under torch.distributed.run: every rank holds the ensemble and takes a contiguous block of the stacks
    ws, rank, local_rank = distributed.init_process_group_from_env()
    handle = _lib.Handle(None, local_rank)
    blobs = []
    for i, idx in enumerate(order):
        print(f"Loading classifier {i}...", flush=True)
        try:
            blobs.append(load_member_weights(best_ensemble_dir, idx, last_resnet_layer))
        except FileNotFoundError as e:
            print(f"{FAIL} {e}", flush=True)
            sys.exit(1)
        print(f"... Classifier {i} loaded.", flush=True)
    ens = inv_depth.InvDepthEnsemble(handle, blobs, size=resnet_inp_shape[0])
    print("All classifiers loaded.", flush=True)
    print(OK, flush=True)

    from compute_branches import n_planes
    test_path = sorted(glob(str(in_root / "*")))[0]
    try:
        if os.path.isdir(test_path) or n_planes(test_path) == 1:
            zstack_paths = zs.find_zstack_image_sequences(str(in_root))
        else:
            zstack_paths = zs.find_zstack_files(str(in_root))
    except zs.ZStackInputException as exc:
        print(f"{FAIL} {exc}")
        sys.exit(1)

    rows = []
    failed = False
    pending = []

    def flush_pending():
        # every step is per slice: stacks of one shape go through the classifiers together (InvDepthEnsemble.predict_stacks)
        for (si_, _), probs in zip(pending, ens.predict_stacks([im for _, im in pending])):     # (Z, n_pred_models): yhatp_m of compute_inv_depth.py:154
            for z, (inv_prob, inv_label) in enumerate(inv_depth.ensemble_predictions(probs, cls_thresh)):
                rows.append((si_ * (1 << 20) + z, inv_label, float(inv_prob), 0.0))       # float32 -> float64 is exact; back below
        pending.clear()

    stack_ids = sorted(zstack_paths)              # a deterministic order the ranks agree on (the reference keeps glob order)
    try:
        for si in distributed.shard_indices(len(stack_ids), rank, ws):
            zstack_id = stack_ids[int(si)]
            zstack_path = zstack_paths[zstack_id]
            print(f"Processing {zstack_id}...", flush=True)
            try:
                if isinstance(zstack_path, str) and zstack_path.endswith(".npy"):
                    img = np.load(zstack_path)
                else:
                    img, _ = helper.load_image(zstack_path, args.time, args.channel)
            except (OSError, ValueError) as error:
                print(f"{FAIL}{error}", flush=True)
                failed = True                         # still enter the gather below: the other ranks are waiting in it
                break
            if img.ndim == 2:
                img = img[None]
            pending.append((int(si), img))
            if sum(len(im) for _, im in pending) >= 128:
                flush_pending()
        if not failed:
            flush_pending()
    except Exception:                                 # noqa: BLE001 -- a library / HIP error must not leave the other ranks in the gather
        import traceback
        traceback.print_exc()
        print(f"{FAIL}rank {rank}: the shard failed", flush=True)
        failed = True

    try:
        gathered = distributed.gather_rows_ragged(rows, failed=failed)
    except distributed.RankFailed:
        handle.close()
        distributed.finish_process_group()
        sys.exit(1)
    rows = [(f"{stack_ids[r[0] >> 20]}_z{r[0] & ((1 << 20) - 1)}", np.float32(r[2]), r[1]) for r in gathered]
    if rank != 0:
        handle.close()
        distributed.finish_process_group()
        return
    print("Saving results...", flush=True)
    out_csv_path = get_unique_output_filepath(out_root / "invasion_depth_predictions.csv")
    with open(out_csv_path, "w", newline="") as f:               # pandas DataFrame.to_csv with the slice ids as the index
        wr = csv.writer(f, lineterminator="\n")
        wr.writerow(["Z Slice ID", "Invasion Probability", "Invasion Prediction (0=no 1=yes)"])
        for sid, p, lab in rows:
            wr.writerow([sid, str(np.float32(p)), lab])           # a float32 column: pandas writes its shortest float32 repr
    print("... Results saved.", flush=True)
    print(OK, flush=True)
    handle.close()
    distributed.finish_process_group()


if __name__ == "__main__":
    main()
