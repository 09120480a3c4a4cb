"""Device PNG decode timings (gpurun): the two kernels over a batch of frame PNGs, and the ingest library's device mode against its host mode.
    python tools/png_bench.py [n_images] [n_files] [levels, e.g. 1,6]"""
import ctypes, io, json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image
from libcimbar_amd import HipDecoder, decoder, framegen, ingest, modeb

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nfiles = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
dev = torch.device("cuda", 0)
dec = HipDecoder(0)
lib = decoder.load_library()
payload = framegen.synth_payload(128, seed=5151, device=dev)
fr = torch.empty((128, modeb.IMG, modeb.IMG, 3), dtype=torch.uint8, device=dev)
dec.encode_batch_device(payload.data_ptr(), 128, fr.data_ptr())
torch.cuda.synchronize()
host128 = fr.cpu().numpy()
out = {}
levels = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "1,6").split(",")]
variant = os.environ.get("CIMBAR_HIP_PNG_SIMT", "0")
writer = os.environ.get("PNG_WRITER", "pillow")        # pillow: adaptive filters; opencv: cv::imwrite's defaults (Sub filter, Z_RLE)
for lvl, n in [(l, n) for l in levels] + [(1, 4 * n)]:
    pngs = []
    for k in range(128):
        if writer == "opencv":
            # what cv::imwrite writes by default, i.e. what the reference's encoder CLI produces (imgcodecs grfmt_png.cpp: PNG_FILTER_SUB on every
            # row, Z_BEST_SPEED, strategy Z_RLE)
            import zlib
            from tests import png_cases
            pngs.append(png_cases.make_png(host128[k], [1] * host128[k].shape[0], level=lvl, strategy=zlib.Z_RLE))
            continue
        buf = io.BytesIO()
        Image.fromarray(host128[k]).save(buf, format="PNG", compress_level=lvl)
        pngs.append(buf.getvalue())
    desc = (decoder.PngDesc * n)()
    blob = bytearray()
    offs = []
    for k in range(128):
        w, h, ct, _d, _i, z, _p = decoder.png_split(pngs[k])
        while len(blob) % 16:
            blob.append(0)
        offs.append((len(blob), len(z)))
        blob += z
    while len(blob) % 16:
        blob.append(0)
    for i in range(n):
        desc[i].zoff, desc[i].zlen = offs[i % 128]
        desc[i].width, desc[i].height, desc[i].color_type = 1024, 1024, 2
    d_z = torch.from_numpy(np.frombuffer(bytes(blob), np.uint8).copy()).to(dev)
    d_desc = torch.from_numpy(np.frombuffer(bytes(desc), np.uint8).copy()).to(dev)
    ss = int(lib.cimbar_hip_png_scratch_bytes(1024, 1024, 2))
    rs = 1024 * 1024 * 3
    d_scratch = torch.empty(n * ss, dtype=torch.uint8, device=dev)
    d_rgb = torch.zeros((n, 1024, 1024, 3), dtype=torch.uint8, device=dev)
    d_status = torch.zeros(n, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream(dev)
    best = None
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = lib.cimbar_hip_png_decode_batch(0, d_z.data_ptr(), d_z.numel(), d_desc.data_ptr(), n, d_scratch.data_ptr(), ss, d_rgb.data_ptr(), rs, d_status.data_ptr(), ctypes.c_void_p(st.cuda_stream))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    ok = bool((d_status == 0).all().item()) and all(bool((d_rgb[i] == fr[i % 128]).all().item()) for i in range(0, n, 37))
    out[f"kernels_level{lvl}_n{n}"] = {"simt": variant, "writer": writer, "images": n, "ms": round(best * 1e3, 2), "images_per_s": round(n / best, 1), "ok": ok, "rc": rc,
                                  "avg_zlib_bytes": int(sum(l for _o, l in offs) / 128)}
    print(json.dumps(out[f"kernels_level{lvl}_n{n}"]), flush=True)
    del d_scratch, d_rgb
    torch.cuda.empty_cache()
# ingest: device mode vs host mode on the same files
with tempfile.TemporaryDirectory() as td:
  if nfiles > 0:
      paths = []
      for k in range(128):
          pth = os.path.join(td, f"f{k:03d}.png")
          Image.fromarray(host128[k]).save(pth, compress_level=1)
          paths.append(pth)
      paths = (paths * ((nfiles + 127) // 128))[:nfiles]
      for label, kw in (("device_b4096r3", dict(batch_frames=4096, ring=3, png_device=True, zbytes_per_frame=360000)), ("device_b2048r3", dict(batch_frames=2048, ring=3, png_device=True, zbytes_per_frame=400000)),
                        ("host_b64", dict(batch_frames=64, ring=3))):
          ing = ingest.Ingest(dec, threads=0, **kw)
          ing.run_files(paths[:min(kw["batch_frames"], 256)])
          t0 = time.perf_counter()
          total, chunks, masks = ing.run_files(paths)
          dt = time.perf_counter() - t0
          tm = ing.timings()
          ok = total == nfiles * 7500 and bool((torch.from_numpy(chunks[:128]) == payload.cpu()).all())
          row = {"files": nfiles, "ms": round(dt * 1e3, 2), "frames_per_s": round(nfiles / dt, 1), "payload_ok": ok, "host_fill_cpu_s": round(tm["host_fill_s"], 3),
                 "device_wait_s": round(tm["device_wait_s"], 3)}
          if "png_device" in kw:
              row.update(ing.png_stats())
          out["ingest_" + label] = row
          print(label, json.dumps(row), flush=True)
          ing.close()
json.dump(out, open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "png_bench.json"), "w"), indent=1)
