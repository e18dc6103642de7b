echo "== probe python packages on the GPU box"; python - <<'PY'
import importlib
for m in ["diffusers","xformers","torchvision","transformers","skimage","cv2","PIL","safetensors","open_clip","timm"]:
    try:
        mod = importlib.import_module(m); print(m, "OK", getattr(mod, "__version__", "?"))
    except Exception as e:
        print(m, "MISSING", type(e).__name__, str(e)[:80])
PY
ls ~/.cache/huggingface 2>/dev/null | head; ls /root/.cache/torch/hub 2>/dev/null | head; nproc; free -g | head -2
