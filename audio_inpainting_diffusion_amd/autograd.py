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
