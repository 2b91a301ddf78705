"""Equal-spp and equal-time RMSE study (second half of the BASELINE metric), shared by tools/rmse_report.py and the C5 full-size test
(tests/test_gpu_configs.py).  RMSE on the linear accum buffer of "pt", "SPCBPT_eye" with the minimal tuple, "SPCBPT_eye" with the
trained tuple and "plain BDPT" (uniformSample, cuProg.h:283-289) at N spp, against a reference of 32 N spp (16 N of "pt" + 16 N of
"SPCBPT_eye", trained) whose subframe indices are disjoint from the images under test, so no samples are shared.
Each frame is launched on a cleared accum buffer at subframe index s (the kernel then leaves new/(s+1) in it) and summed on the
device in fp32 chunks / fp64 totals, which gives a plain mean for any set of indices.
Equal time: every algorithm's frame time is measured in the pipelined loop bench.py uses (no sync between frames); "pt" and plain BDPT
are then rendered AGAIN with the number of samples they afford in the time N samples of the trained SPCBPT take."""
import time

import numpy as np


def rmse_study(p, scene, W, H, N, train_paths, light_paths=100000, log=None):
    """p: the package; returns the report dict.  The renderer must have been created with SPCBPT_EYE_BATCH >= 4 in the environment."""
    import torch
    r = p.Renderer(scene, 0)
    c = scene.camera
    r.set_camera_lookat(c["eye"], c["lookat"], c["up"], c["fov"], W / H)
    r.resize(W, H)
    r.set_light_trace(light_paths, 52, 1)
    view = p.dist.device_view(r.accum_device_ptr(), W * H * 16, torch.device("cuda:0")).view(torch.float32).view(H, W, 4)
    finite = {}

    def mean_image(alg, first, n, key=None):
        tot = torch.zeros(H, W, 3, dtype=torch.float64, device="cuda:0")
        for s in range(first, first + n):
            r.clear_accum()
            r.render_frame(alg, s, launch_frame=s + 1)
            r.sync()
            tot += view[..., :3].double() * (s + 1)
            torch.cuda.synchronize()
        if key:
            finite[key] = bool(torch.isfinite(tot).all().item()) and bool((view[..., 3] == 1.0).all().item())
        return (tot / n).cpu().numpy()

    def frame_ms(alg, frames=24):
        """ms per frame in the host loop bench.py runs: "pt" frame by frame; "SPCBPT_eye" with the light pass launched a batch ahead
        and 4 frames per eye launch"""
        def loop(n):
            if alg != "SPCBPT_eye":
                for s in range(n): r.render_frame(alg, s)
                return
            r.set_light_ahead(True)
            nxt = 1
            for _ in range(4):
                r.launch("light trace", nxt); nxt += 1
            queued = []
            for s in range(n):
                r.launch("light trace", nxt); nxt += 1
                r.build_sampler()
                queued.append(s)
                if len(queued) == 4 or s == n - 1:
                    r.launch_eye_batch(queued); queued = []
            r.sync()
            r.set_light_ahead(False)
        loop(4)
        r.sync(); torch.cuda.synchronize()
        t = time.perf_counter()
        loop(frames)
        r.sync(); torch.cuda.synchronize()
        return (time.perf_counter() - t) / frames * 1e3

    out = {"scene": scene.name, "width": W, "height": H, "spp": N, "ref_spp": 32 * N, "triangles": int(len(scene.indices))}
    t0 = time.time()
    r.set_subspace()                                   # minimal tuple
    pt = mean_image("pt", 0, N, "pt")
    sp_min = mean_image("SPCBPT_eye", 0, N, "spcbpt_minimal")
    ms_pt, ms_min = frame_ms("pt"), frame_ms("SPCBPT_eye")
    pt_ref = mean_image("pt", 4 * N, 16 * N)
    t1 = time.time(); r.preprocess(train_paths, train_paths, True); out["preprocess_seconds"] = time.time() - t1
    et, lt, q, cmf = r.get_subspace()
    out["eye_subspaces"] = int(len(set(et["label"][et["leaf"] == 1].tolist())))
    out["light_subspaces"] = int(len(set(lt["label"][lt["leaf"] == 1].tolist())))
    sp_tr = mean_image("SPCBPT_eye", 0, N, "spcbpt_trained")
    ms_tr = frame_ms("SPCBPT_eye")
    # "plain BDPT", the comparator BASELINE config 5 names: the same kernel with SubspaceSampler_device::uniformSample (cuProg.h:283-289)
    # in place of the two-stage subspace sampler (spcbpt_set_connection_sampler), at N spp and at the spp it affords in the same time
    r.set_connection_sampler(1)
    bd = mean_image("SPCBPT_eye", 0, N, "plain_bdpt")
    ms_bd = frame_ms("SPCBPT_eye")
    n_bd = max(1, min(4 * N - 1, int(round(N * ms_tr / ms_bd))))
    bd_eq = mean_image("SPCBPT_eye", 0, n_bd)
    r.set_connection_sampler(0)
    n_eq = max(1, min(4 * N - 1, int(round(N * ms_tr / ms_pt))))   # samples "pt" affords in the time of N trained-SPCBPT samples
    pt_eq = mean_image("pt", 0, n_eq)
    sp_ref = mean_image("SPCBPT_eye", 32 * N, 16 * N)
    ref = 0.5 * (pt_ref + sp_ref)
    rm = lambda a: float(np.sqrt(((a - ref) ** 2).mean()))
    rel = lambda a: float((((a - ref) ** 2) / (ref ** 2 + 1e-2)).mean())
    out.update(rmse_pt=rm(pt), rmse_spcbpt_minimal=rm(sp_min), rmse_spcbpt_trained=rm(sp_tr), rmse_plain_bdpt=rm(bd),
               relmse_pt=rel(pt), relmse_spcbpt_minimal=rel(sp_min), relmse_spcbpt_trained=rel(sp_tr), relmse_plain_bdpt=rel(bd),
               mean_pt_ref=float(pt_ref.mean()), mean_spcbpt_ref=float(sp_ref.mean()),
               mean_pt=float(pt.mean()), mean_spcbpt_trained=float(sp_tr.mean()), mean_plain_bdpt=float(bd.mean()),
               ref_disagreement_rmse=float(np.sqrt(((pt_ref - sp_ref) ** 2).mean())), every_pixel_written_and_finite=finite,
               seconds=time.time() - t0)
    out["variance_ratio_pt_over_trained"] = (out["rmse_pt"] / out["rmse_spcbpt_trained"]) ** 2
    out["variance_ratio_plain_bdpt_over_trained"] = (out["rmse_plain_bdpt"] / out["rmse_spcbpt_trained"]) ** 2
    out["ms_per_frame"] = {"pt": ms_pt, "spcbpt_minimal": ms_min, "spcbpt_trained": ms_tr, "plain_bdpt_uniformSample": ms_bd}
    out["equal_time"] = {"budget_ms": N * ms_tr, "spp": {"pt": n_eq, "spcbpt_trained": N, "plain_bdpt": n_bd},
                         "rmse_plain_bdpt_measured": rm(bd_eq), "relmse_plain_bdpt_measured": rel(bd_eq),
                         "variance_ratio_plain_bdpt_over_trained": (rm(bd_eq) / rm(sp_tr)) ** 2,
                         "relmse_ratio_plain_bdpt_over_trained": rel(bd_eq) / rel(sp_tr),
                         "rmse_pt_measured": rm(pt_eq), "relmse_pt_measured": rel(pt_eq),
                         "rmse_pt_derived": out["rmse_pt"] * (ms_pt / ms_tr) ** 0.5,
                         "rmse_spcbpt_minimal_derived": out["rmse_spcbpt_minimal"] * (ms_min / ms_tr) ** 0.5,
                         "rmse_spcbpt_trained": out["rmse_spcbpt_trained"], "relmse_spcbpt_trained": out["relmse_spcbpt_trained"]}
    out["equal_time"]["variance_ratio_pt_over_trained"] = (out["equal_time"]["rmse_pt_measured"] / out["rmse_spcbpt_trained"]) ** 2
    r.close()
    return out
