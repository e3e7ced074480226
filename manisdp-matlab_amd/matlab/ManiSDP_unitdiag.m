function [X, obj, data] = ManiSDP_unitdiag(At, b, c, K, options)
%MANISDP_UNITDIAG  GPU drop-in for the reference's src/primal/ManiSDP_unitdiag.m:
%   Min <C, X>  s.t.  A(X) = b,  X >= 0,  X_ii = 1   (SeDuMi data At, b, c, K.s = n).
%   Same call, option names, defaults, printed lines and data fields (including data.fac_size); the work is
%   done by msdp_al_engine over libmanisdp_hip (see that file).
if nargin < 5, options = struct(); end
defaults = {'p0', 2; 'AL_maxiter', 300; 'gama', 2; 'sigma0', 1e-3; 'sigma_min', 1e-2; 'sigma_max', 1e7; ...
            'tol', 1e-8; 'theta', 1e-3; 'delta', 8; 'alpha', 0.1; 'tolgradnorm', 1e-8; ...
            'TR_maxinner', 20; 'TR_maxiter', 4; 'tau1', 1; 'tau2', 1; 'line_search', 0};
[X, obj, data] = msdp_al_engine('unitdiag', struct('n', K.s, 'At', At, 'b', b, 'c', c), options, defaults);
end
