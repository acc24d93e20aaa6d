"""torch.autograd bridge: lets ``torch.autograd.grad(norm, x)`` (reference testing/edm_sampler_inpainting.py:78)
flow through the MI355X network when the REFERENCE's own sampler drives it.  The backward pass is the
hand-written input-VJP launch plan (network._body_vjp), not torch autograd over eager ops."""
import torch


class DenoiserFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, sigma, net):
        ctx.net = net
        ctx.B = inputs.shape[0]
        ctx.stamp = net._fwd_stamp = getattr(net, "_fwd_stamp", 0) + 1
        return net._forward_impl(inputs, sigma)

    @staticmethod
    def backward(ctx, g):
        net = ctx.net
        if net._fwd_stamp != ctx.stamp:
            raise RuntimeError("the network was evaluated again before backward(): saved activations were overwritten "
                               "(the launch plan keeps ONE set of activations per batch size)")
        return net.vjp(g), None, None


class TrainFn(torch.autograd.Function):
    """Forward of the network in training mode, differentiable w.r.t. its PARAMETERS (and the input): lets the reference's own
    training step -- ``error, sigma = diff_params.loss_fn(network, audio); error.mean().backward(); optimizer.step()``
    (training/trainer.py:262-281) -- run unchanged on the MI355X network.  backward() seeds the hand-written backward plan with
    the incoming gradient and hands the parameter gradients it writes (csrc/aid_train.hip) back to torch.autograd."""

    @staticmethod
    def forward(ctx, inputs, sigma, net, *params):
        ctx.net = net
        ctx.stamp = net._fwd_stamp = getattr(net, "_fwd_stamp", 0) + 1
        ctx.need_input = inputs.requires_grad
        return net._train_forward(inputs, sigma)

    @staticmethod
    def backward(ctx, g):
        net = ctx.net
        if net._fwd_stamp != ctx.stamp:
            raise RuntimeError("the network was evaluated again before backward(): saved activations were overwritten")
        gin, pgrads = net._train_backward(g, need_input=ctx.need_input)
        return (gin, None, None) + tuple(pgrads)
