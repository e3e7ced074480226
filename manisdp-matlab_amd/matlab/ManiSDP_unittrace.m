function [X, obj, data] = ManiSDP_unittrace(At, b, c, K, options)
%MANISDP_UNITTRACE  GPU drop-in for the reference's src/primal/ManiSDP_unittrace.m:
%   Min <C, X>  s.t.  A(X) = b,  X >= 0,  tr(X) = 1   (SeDuMi data At, b, c, K.s = n; factor Y is n x p).
%   Same call, option names, defaults (note sigma_min > sigma0, as in the reference), printed lines and data
%   fields; options.Y0 is honoured as in the reference.  The work is done by msdp_al_engine over libmanisdp_hip.
if nargin < 5, options = struct(); end
defaults = {'p0', 1; 'AL_maxiter', 1000; 'gama', 2; 'sigma0', 1e1; 'sigma_min', 1e2; 'sigma_max', 1e7; ...
            'tol', 1e-8; 'theta', 1e-2; 'delta', 8; 'alpha', 0.05; 'tolgradnorm', 1e-8; ...
            'TR_maxinner', 40; 'TR_maxiter', 3; 'tau1', 1e-5; 'tau2', 1e-4; 'line_search', 1};
[X, obj, data] = msdp_al_engine('unittrace', struct('n', K.s, 'At', At, 'b', b, 'c', c), options, defaults);
end
