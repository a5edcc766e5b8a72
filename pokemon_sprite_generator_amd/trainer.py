"""ImprovedDiffusionTrainer mirror (reference: src/training/improved_diffusion_trainer.py:77-692).

Keeps the reference's surface — `__init__(config, vae_checkpoint_path, experiment_name)`,
`train()`, `train_epoch`, `validate_epoch`, `ddpm_sample`, `generate_samples`,
`save_checkpoint`, `load_checkpoint` — so `train_3stage.py:130-140` runs unchanged
when this class is imported in place of the reference's, and adds the facade
the north-star names: `train_step(latents, text_emb, t, noise=None)` (the loop
body :363-413) and `sample(text_emb, num_samples, fast_sampling, noise_fn=None)`
(:508-569).

Out-of-scope collaborators (BERT text encoder, frozen VAE, data loaders,
TensorBoard) are NOT re-implemented: they are taken from the reference package
(`src.models`, `src.data`) when it is importable, or injected through
`components=` (tests, benchmarks with synthetic latents).
"""
import logging
import os
from pathlib import Path
from typing import Any, Dict, Optional

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr
from .ddp import BucketedAllReduce
from .optim import FusedAdamW, GradArena, ParamArena
from .scheduler import NoiseScheduler
from .unet import UNet


class _NullWriter:
    def add_scalar(self, *a, **k):
        pass

    def add_image(self, *a, **k):
        pass

    def close(self):
        pass


class DiffusionStepper:
    """The accelerated hot path: add_noise -> U-Net -> SmoothL1 -> backward -> [all-reduce] -> clip -> AdamW.

    Everything stays on the device; the only host read is the caller's choice
    (`.item()` on the returned tensors).  The reference's 478 per-parameter
    `.item()` syncs (:399-404) and 5 full-tensor NaN scans (:353-393) are
    replaced by one fused norm reduction and a device-side flag word.
    """

    def __init__(self, unet: UNet, noise_scheduler: NoiseScheduler, lr=1e-4, betas=(0.9, 0.999), weight_decay=0.01,
                 eps=1e-6, max_grad_norm=1.0, optimizer_type="adamw", distributed=None, bucket_bytes=64 << 20):
        self.unet, self.noise_scheduler = unet, noise_scheduler
        self.max_grad_norm = max_grad_norm
        self.device = next(unet.parameters()).device
        if self.device.type != "cuda":
            raise _lib.PsgError("DiffusionStepper needs the U-Net on a GPU (HIP path only)")
        self.lib = _lib.init(self.device.index if self.device.index is not None else torch.cuda.current_device())
        self.reducer = None
        self.params = ParamArena(unet.parameters())        # flat fp32 masters, conv weights OHWI
        if getattr(unet, "compute_dtype", None) == torch.bfloat16 and optimizer_type == "adamw" and not os.environ.get("PSG_NO_SHADOW"):
            self.params.enable_shadow()                     # AdamW also emits next step's bf16 forward weights
        self.arena = GradArena(unet.parameters(), on_ready=lambda i: self.reducer.on_ready(i) if self.reducer is not None else None)
        # Adam (non-decoupled decay) is torch's when asked for (:285-291); AdamW is the fused kernel (:277-283)
        if optimizer_type == "adamw":
            self.optimizer = FusedAdamW(self.arena.params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                        param_arena=self.params, grad_arena=self.arena)
        else:
            self.optimizer = torch.optim.Adam(self.arena.params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        if distributed is None:
            distributed = torch.distributed.is_available() and torch.distributed.is_initialized()
        self.reducer = BucketedAllReduce(self.arena.flat, self.arena.params, self.arena.offsets, bucket_bytes) if distributed else None
        self.flag = torch.zeros(1, dtype=torch.int32, device=self.device)      # bit0 noisy, bit2 eps_hat, bit3 loss non-finite
        self.loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.criterion_beta = 0.1                                               # nn.SmoothL1Loss(beta=0.1) :300

    # ---- fused SmoothL1 + dL/d(eps_hat) ------------------------------------------------------------
    def smooth_l1(self, pred, target, want_grad=True):
        pred_c, tgt = pred.detach().contiguous(), target.detach().contiguous().float()
        grad = torch.empty_like(pred_c) if want_grad else None
        ws = _lib.workspace(self.lib.psg_reduce_workspace_bytes(), self.device)
        check(self.lib.psg_smooth_l1_f32(ptr(pred_c), ptr(tgt), ptr(grad), ptr(self.loss), ptr(self.flag), self.criterion_beta, 1.0,
                                         pred_c.numel(), ptr(ws), stream_ptr()), "psg_smooth_l1_f32")
        return self.loss, grad

    def train_step(self, latents, text_emb, t, noise=None, lr=None) -> Dict[str, torch.Tensor]:
        """One optimizer step on a batch of clean latents (the body of train_epoch, :363-413).

        latents [B,8,27,27] fp32 (un-clamped VAE output), text_emb [B,S,text_dim], t [B] int64, optional
        noise (default randn_like).  Returns device tensors {'loss','grad_norm','nan_flag'} (no sync)."""
        self.unet.train()
        if noise is None:
            noise = torch.randn_like(latents)
        self.flag.zero_()
        noisy = self.noise_scheduler.add_noise(latents, noise, t, clamp=True, flag=self.flag)     # :363, :374
        self.arena.zero()                                                                           # :380
        eps_hat = self.unet(noisy, t, text_emb)                                                     # :381
        loss, dpred = self.smooth_l1(eps_hat, noise)                                                # :388
        eps_hat.backward(dpred)                                                                     # :396
        self.arena.finalize()                      # (joins the weight-gradient side stream)
        if self.reducer is not None:
            self.reducer.finish()
            torch.distributed.all_reduce(self.flag, op=torch.distributed.ReduceOp.MAX)              # all ranks skip together
        normsq = self.arena.grad_norm_sq()                                                          # :399-404
        if lr is not None:
            for g in self.optimizer.param_groups:
                g["lr"] = lr
        if isinstance(self.optimizer, FusedAdamW):
            self.optimizer.step(normsq=normsq, max_norm=self.max_grad_norm, skip_flag=self.flag)    # :410-412
        else:
            check(self.lib.psg_clip_scale_f32(ptr(self.arena.flat), self.arena.numel, ptr(normsq), float(self.max_grad_norm),
                                              stream_ptr()), "psg_clip_scale_f32")
            self.optimizer.step()
            from .ops import WeightCache
            WeightCache.invalidate()
        return {"loss": loss.clone(), "grad_norm": normsq.sqrt(), "nan_flag": self.flag.clone()}

    @torch.no_grad()
    def eval_loss(self, latents, text_emb, t, noise=None):
        """validate_epoch body (:465-486)."""
        self.unet.eval()
        if noise is None:
            noise = torch.randn_like(latents)
        self.flag.zero_()
        noisy = self.noise_scheduler.add_noise(latents, noise, t, clamp=True, flag=self.flag)
        eps_hat = self.unet(noisy, t, text_emb)
        loss, _ = self.smooth_l1(eps_hat, noise, want_grad=False)
        return loss.clone(), self.flag.clone()

    @torch.no_grad()
    def sample(self, text_emb, num_samples, fast_sampling=True, noise_fn=None, latent_dim=8, hw=27, trace=None, use_graph=None):
        """ddpm_sample (:508-569): x <- (x - c2*eps)/sqrt(alpha_t) [+ sqrt(beta_t)*z if t>0], t strided by 50 when fast.
        noise_fn(i, shape) -> tensor supplies x_T (i = -1) and the per-step z (tests inject it); default torch.randn.

        use_graph=True (or PSG_GRAPH=1) captures the loop body (U-Net forward + update, ~700 launches) ONCE into a
        hipGraph (torch.cuda.CUDAGraph over the library's launches on the capture stream) and replays it per step:
        the timestep lives in device memory (the embedding and the update kernel both read it there) and z is
        refilled in place, so the captured work is identical for every step; results are bit-identical to the eager
        path (tests/test_unet_gpu.py).  It is opt-in because it does not pay on MI355X at sampling batch sizes
        (tools/sampler_bench.py: 8.2 vs 7.8 ms/step at B=1..16): the step is bound by the GPU-side latency of ~700
        small dependent kernels, not by host launch cost.  A capture failure falls back to eager with a warning."""
        self.unet.eval()
        dev = self.device
        rnd = noise_fn if noise_fn is not None else (lambda i, shape: torch.randn(shape, device=dev))
        x = rnd(-1, (num_samples, latent_dim, hw, hw)).to(dev).float().contiguous()
        sch = self.noise_scheduler.to(dev)
        c1, c2, sg = sch.step_tables(dev)
        T = sch.num_timesteps
        steps = list(range(0, T, 50)) if fast_sampling else list(range(T))
        t_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        tv = torch.zeros((num_samples,), device=dev, dtype=torch.long)
        z = torch.zeros_like(x)
        text_emb = text_emb.to(dev)
        if use_graph is None:
            use_graph = bool(os.environ.get("PSG_GRAPH"))

        def body():
            eps = self.unet(x, tv, text_emb).contiguous()
            check(self.lib.psg_ddpm_update_f32(ptr(x), ptr(eps), ptr(z), ptr(c1), ptr(c2), ptr(sg), ptr(t_dev), x.numel(), stream_ptr()),
                  "psg_ddpm_update_f32")

        graph = None
        order = list(reversed(steps))
        first = 0
        if use_graph and len(order) > 2:
            try:
                # step 0 runs eagerly: it sizes the workspace and fills the prepared-weight cache (no allocation and
                # no weight preparation may happen inside the capture), and it is a real step of the chain
                self._sample_set(order[0], 0, tv, t_dev, z, rnd, x)
                body()
                if trace is not None:
                    trace.append(x.clone())
                first = 1
                torch.cuda.synchronize(dev)
                graph = torch.cuda.CUDAGraph()
                self._sample_set(order[1], 1, tv, t_dev, z, rnd, x)
                x_before = x.clone()
                with torch.cuda.graph(graph):
                    body()
                x.copy_(x_before)                       # capture does not execute: the replay below runs step 1
            except Exception as e:                      # noqa: BLE001
                logging.getLogger(__name__).warning(f"hipGraph capture of the sampler step failed ({e!r}); running eagerly")
                graph = None
        for i in range(first, len(order)):
            self._sample_set(order[i], i, tv, t_dev, z, rnd, x)
            if graph is not None:
                graph.replay()
            else:
                body()
            if trace is not None:
                trace.append(x.clone())
        return x

    @staticmethod
    def _sample_set(t, i, tv, t_dev, z, rnd, x):
        """Per-step inputs, written in place (the captured graph reads these buffers)."""
        tv.fill_(t)
        t_dev.fill_(t)
        if t > 0:
            z.copy_(rnd(i, tuple(x.shape)).to(device=x.device, dtype=torch.float32))


class ImprovedDiffusionTrainer:
    """Drop-in for the reference class of the same name (stage-2 diffusion trainer)."""

    def __init__(self, config: Dict[str, Any], vae_checkpoint_path: str, experiment_name: str = "pokemon_diffusion",
                 components: Optional[Dict[str, Any]] = None, compute_dtype: Optional[torch.dtype] = None):
        self.config, self.vae_checkpoint_path, self.experiment_name = config, vae_checkpoint_path, experiment_name
        self._components = components or {}
        mi = config.get("mi355x", {}) or {}
        if compute_dtype is None:
            compute_dtype = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp32": torch.float32,
                             "float32": torch.float32}[str(mi.get("dtype", "bf16"))]
        self.compute_dtype = compute_dtype
        if not torch.cuda.is_available():
            raise _lib.PsgError("ImprovedDiffusionTrainer (MI355X build) needs a GPU; there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device())
        print(f"Using device: {self.device}")
        self.setup_directories()
        self.setup_logging()
        self.setup_models()
        self.setup_data_loaders()
        self.setup_optimization()
        self.setup_scheduler()
        self.setup_monitoring()
        self.current_epoch, self.global_step, self.best_val_loss = 0, 0, float("inf")

    # -- setup (plain Python glue, same keys as the reference) -----------------------------------------
    def setup_directories(self):
        self.experiment_dir = Path(self.config["experiment_dir"]) / self.experiment_name
        self.checkpoint_dir, self.log_dir, self.sample_dir = (self.experiment_dir / d for d in ("checkpoints", "logs", "samples"))
        for d in (self.experiment_dir, self.checkpoint_dir, self.log_dir, self.sample_dir):
            d.mkdir(parents=True, exist_ok=True)

    def setup_logging(self):
        logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s",
                            handlers=[logging.FileHandler(self.log_dir / "diffusion_training.log"), logging.StreamHandler()])
        self.logger = logging.getLogger(__name__)

    def _component(self, name):
        if name in self._components:
            return self._components[name]
        try:                                   # the reference package, when this class is dropped into its tree
            import importlib
            mod = importlib.import_module({"create_data_loaders": "src.data"}.get(name, "src.models"))
            return getattr(mod, name)
        except Exception as e:                 # noqa: BLE001
            raise _lib.PsgError(f"component '{name}' is outside the accelerated hot path: provide it via components= "
                                f"or run inside the reference tree ({e})")

    def setup_models(self):
        mc = self.config["model"]
        latent_dim = mc.get("latent_dim", 8)
        if "text_encoder" in self._components:
            self.text_encoder = self._components["text_encoder"]
            self.vae_encoder = self._components["vae_encoder"]
            self.vae_decoder = self._components.get("vae_decoder")
        else:
            TextEncoder, VAEEncoder, VAEDecoder = (self._component(n) for n in ("TextEncoder", "VAEEncoder", "VAEDecoder"))
            self.text_encoder = TextEncoder(model_name=mc["bert_model"], hidden_dim=mc["text_embedding_dim"]).to(self.device)
            print(f"Loading VAE from {self.vae_checkpoint_path}")
            ckpt = torch.load(self.vae_checkpoint_path, map_location=self.device)
            self.vae_encoder = VAEEncoder(input_channels=3, latent_dim=latent_dim).to(self.device)
            self.vae_decoder = VAEDecoder(latent_dim=latent_dim, text_dim=mc["text_embedding_dim"], output_channels=3).to(self.device)
            if "vae_state_dict" in ckpt:
                enc = {k[8:]: v for k, v in ckpt["vae_state_dict"].items() if k.startswith("encoder.")}
                dec = {k[8:]: v for k, v in ckpt["vae_state_dict"].items() if k.startswith("decoder.")}
                self.vae_encoder.load_state_dict(enc, strict=False)
                self.vae_decoder.load_state_dict(dec, strict=False)
            if "text_encoder_state_dict" in ckpt:
                self.text_encoder.load_state_dict(ckpt["text_encoder_state_dict"], strict=False)
        for m in (self.text_encoder, self.vae_encoder, self.vae_decoder):
            if isinstance(m, torch.nn.Module):
                for p in m.parameters():
                    p.requires_grad = False
                m.eval()
        self.unet = UNet(latent_dim=latent_dim, text_dim=mc["text_embedding_dim"], time_emb_dim=mc.get("time_emb_dim", 128),
                         num_heads=mc.get("num_heads", 4), compute_dtype=self.compute_dtype).to(self.device)
        self.noise_scheduler = NoiseScheduler(num_timesteps=mc.get("num_timesteps", 1000), beta_start=mc.get("beta_start", 0.0001),
                                              beta_end=mc.get("beta_end", 0.02))
        self.logger.info(f"U-Net initialized with {sum(p.numel() for p in self.unet.parameters())} parameters")

    def setup_data_loaders(self):
        if "data_loaders" in self._components:
            self.data_loaders = self._components["data_loaders"]
            return
        dc, uc = self.config["data"], self.config.get("unet_optimization", {}) or {}
        bs, nw = uc.get("batch_size", dc["batch_size"]), uc.get("num_workers", dc["num_workers"])
        tr, va, te = self._component("create_data_loaders")(csv_path=dc["csv_path"], image_dir=dc["image_dir"], batch_size=bs,
                                                            val_split=dc["val_split"], test_split=dc["test_split"], image_size=dc["image_size"],
                                                            num_workers=nw, pin_memory=dc["pin_memory"])
        self.data_loaders = {"train": tr, "val": va, "test": te}
        self.logger.info(f"Data loaders created: train={len(tr)}, val={len(va)}, test={len(te)}")

    def setup_optimization(self):
        uc, oc = self.config.get("unet_optimization", {}) or {}, self.config["optimization"]
        get = lambda k, default=None: uc.get(k, oc.get(k, default))      # safe fallbacks (the reference KeyErrors on beta1/beta2)
        lr, self.max_grad_norm = get("learning_rate"), get("max_grad_norm")
        self.stepper = DiffusionStepper(self.unet, self.noise_scheduler, lr=lr, betas=(get("beta1", 0.9), get("beta2", 0.999)),
                                        weight_decay=get("weight_decay"), eps=1e-6, max_grad_norm=self.max_grad_norm,
                                        optimizer_type=get("optimizer", "adamw"))
        self.optimizer = self.stepper.optimizer
        self.scheduler_config = {"type": uc.get("scheduler", oc.get("scheduler", "cosine")), "lr": lr}
        self.logger.info(f"Using U-Net optimization: {get('optimizer', 'adamw')}, lr={lr}, wd={get('weight_decay')}, clip={self.max_grad_norm}")

    def setup_scheduler(self):
        if self.scheduler_config["type"] == "cosine":
            total = self.config["training"]["diffusion_epochs"] * len(self.data_loaders["train"])
            self.scheduler = torch.optim.lr_scheduler.OneCycleLR(self.optimizer, max_lr=self.scheduler_config["lr"], total_steps=total,
                                                                 pct_start=0.1, anneal_strategy="cos")
        else:
            self.scheduler = torch.optim.lr_scheduler.ConstantLR(self.optimizer, factor=1.0)

    def setup_monitoring(self):
        try:
            from torch.utils.tensorboard import SummaryWriter
            self.writer = SummaryWriter(log_dir=self.log_dir)
        except Exception:                      # tensorboard is optional here
            self.writer = _NullWriter()

    # -- facade ---------------------------------------------------------------------------------------
    def train_step(self, latents, text_emb, t, noise=None):
        out = self.stepper.train_step(latents, text_emb, t, noise)
        self.scheduler.step()
        self.global_step += 1
        return out

    def sample(self, text_emb, num_samples, fast_sampling=True, noise_fn=None):
        return self.stepper.sample(text_emb, num_samples, fast_sampling, noise_fn, latent_dim=self.config["model"].get("latent_dim", 8))

    def ddpm_sample(self, text_emb: torch.Tensor, num_samples: int, fast_sampling: bool = True) -> torch.Tensor:
        return self.sample(text_emb, num_samples, fast_sampling)

    # -- epochs ----------------------------------------------------------------------------------------
    def _encode(self, batch):
        images = batch["image"].to(self.device)
        with torch.no_grad():
            text_emb = self.text_encoder(batch["full_description"])
            latent = self.vae_encoder(images)
            latent = latent[0] if isinstance(latent, (tuple, list)) else latent
        return latent.float(), text_emb.float()

    def train_epoch(self, epoch: int) -> Dict[str, float]:
        total, nb, nan_count = torch.zeros(1, device=self.device), 0, 0
        log_every = self.config["training"]["log_every"]
        for batch_idx, batch in enumerate(self.data_loaders["train"]):
            try:
                latent, text_emb = self._encode(batch)
                t = torch.randint(0, self.noise_scheduler.num_timesteps, (latent.shape[0],), device=self.device)
                out = self.train_step(latent, text_emb, t)
                if batch_idx % log_every == 0:                      # the only host syncs of the loop
                    if int(out["nan_flag"].item()) != 0:
                        nan_count += 1
                        self.logger.warning("NaN/Inf detected, batch skipped")
                        continue
                    self.writer.add_scalar("Diffusion Train/Loss", out["loss"].item(), self.global_step)
                    self.writer.add_scalar("Diffusion Train/Learning_Rate", self.optimizer.param_groups[0]["lr"], self.global_step)
                    self.writer.add_scalar("Diffusion Train/Gradient_Norm", out["grad_norm"].item(), self.global_step)
                total += torch.where(out["nan_flag"] == 0, out["loss"], torch.zeros_like(out["loss"]))
                nb += 1
            except Exception as e:                                 # noqa: BLE001  (reference :433-435 logs and continues)
                self.logger.error(f"Error in training batch {batch_idx}: {e}")
                continue
        if nb == 0:
            self.logger.error("No valid batches processed!")
            return {"train_loss": float("inf")}
        avg = float(total.item()) / nb
        self.logger.info(f"Epoch {epoch}: Average loss = {avg:.6f}, NaN batches = {nan_count}, LR = {self.optimizer.param_groups[0]['lr']:.2e}")
        return {"train_loss": avg}

    def validate_epoch(self, epoch: int) -> Dict[str, float]:
        total, nb = 0.0, 0
        for batch in self.data_loaders["val"]:
            try:
                latent, text_emb = self._encode(batch)
                t = torch.randint(0, self.noise_scheduler.num_timesteps, (latent.shape[0],), device=self.device)
                loss, flag = self.stepper.eval_loss(latent, text_emb, t)
                if int(flag.item()) == 0:
                    total += float(loss.item())
                    nb += 1
            except Exception as e:                                 # noqa: BLE001
                self.logger.error(f"Error in validation batch: {e}")
        if nb == 0:
            return {"val_loss": float("inf")}
        self.writer.add_scalar("Diffusion Val/Loss", total / nb, epoch)
        return {"val_loss": total / nb}

    def generate_samples(self, epoch: int, num_samples: int = 8):
        """Monitoring only (:571-615): needs the (out-of-scope) VAE decoder; skipped when it is absent."""
        if self.vae_decoder is None:
            return
        batch = next(iter(self.data_loaders["val"]))
        desc = batch["full_description"][:num_samples]
        with torch.no_grad():
            text_emb = self.text_encoder(desc).float()
            for i in range(0, len(desc), 4):
                te = text_emb[i:i + 4]
                imgs = self.vae_decoder(self.ddpm_sample(te, te.shape[0]), te)
                imgs = torch.clamp((imgs + 1.0) / 2.0, 0, 1)
                for j, img in enumerate(imgs):
                    self.writer.add_image(f"Diffusion Generated/Sample_{i + j}", img.cpu(), epoch)
        self.logger.info(f"Generated {len(desc)} samples for epoch {epoch}")

    # -- checkpoints (wire format of :617-655) ------------------------------------------------------------
    def save_checkpoint(self, epoch: int, is_best: bool = False):
        ckpt = {"epoch": epoch, "global_step": self.global_step, "unet_state_dict": self.unet.state_dict(),
                "optimizer_state_dict": self.optimizer.state_dict(), "scheduler_state_dict": self.scheduler.state_dict(),
                "best_val_loss": self.best_val_loss, "config": self.config}
        if is_best:
            torch.save(ckpt, self.checkpoint_dir / "diffusion_best_model.pth")
            self.logger.info(f"New best model saved at epoch {epoch}")

    def load_checkpoint(self, checkpoint_path: str):
        ckpt = torch.load(checkpoint_path, map_location=self.device)
        self.current_epoch, self.global_step, self.best_val_loss = ckpt["epoch"], ckpt["global_step"], ckpt["best_val_loss"]
        self.unet.load_state_dict(ckpt["unet_state_dict"])
        self.optimizer.load_state_dict(ckpt["optimizer_state_dict"])
        if self.scheduler and ckpt["scheduler_state_dict"]:
            self.scheduler.load_state_dict(ckpt["scheduler_state_dict"])
        from .ops import WeightCache
        WeightCache.invalidate()
        self.logger.info(f"Checkpoint loaded from {checkpoint_path}")

    def train(self):
        self.logger.info("Starting diffusion training...")
        tc = self.config["training"]
        for epoch in range(self.current_epoch, tc["diffusion_epochs"]):
            self.current_epoch = epoch
            tm = self.train_epoch(epoch)
            if tm["train_loss"] == float("inf"):
                self.logger.error(f"Training failed at epoch {epoch}, stopping")
                break
            vm = self.validate_epoch(epoch)
            if epoch % tc["sample_every"] == 0:
                self.generate_samples(epoch)
            is_best = vm["val_loss"] < self.best_val_loss
            if is_best:
                self.best_val_loss = vm["val_loss"]
            if epoch % tc["save_every"] == 0 or is_best:
                self.save_checkpoint(epoch, is_best)
            self.logger.info(f"Epoch {epoch}: train_loss={tm['train_loss']:.4f}, val_loss={vm['val_loss']:.4f}")
        self.logger.info("diffusion training completed!")
        self.writer.close()


DiffusionTrainer = ImprovedDiffusionTrainer     # src/training/__init__.py:7 and train_3stage.py:19 alias it this way
