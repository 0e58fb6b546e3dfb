// compile-hygiene stand-in (tests/adapter_stubs/README.md): core/robust_kernel.h:72, core/robust_kernel_impl.h:41-170
#pragma once
namespace g2o {
class RobustKernel { public: virtual ~RobustKernel(); double delta() const; };
class RobustKernelScaleDelta : public RobustKernel {};
class RobustKernelHuber : public RobustKernel {};
class RobustKernelTukey : public RobustKernel {};
class RobustKernelPseudoHuber : public RobustKernel {};
class RobustKernelCauchy : public RobustKernel {};
class RobustKernelSaturated : public RobustKernel {};
class RobustKernelDCS : public RobustKernel {};
}  // namespace g2o
