function [X, obj, data] = ManiSDP(At, b, c, K, options)
%MANISDP  GPU drop-in for the reference's generic entry point src/primal/ManiSDP.m:
%   Min <C, X>  s.t.  A(X) = b,  X >= 0   (Euclidean manifold; factor Y is n x p).
%   Same call, option names, defaults, printed lines and data fields; the work is done by msdp_al_engine over
%   libmanisdp_hip (see that file).
if nargin < 5, options = struct(); end
defaults = {'p0', 1; 'AL_maxiter', 1000; 'gama', 2; 'sigma0', 1e-2; 'sigma_min', 1e-1; 'sigma_max', 1e7; ...
            'tol', 1e-8; 'theta', 1e-2; 'delta', 8; 'alpha', 0.1; 'tolgradnorm', 1e-8; ...
            'TR_maxinner', 20; 'TR_maxiter', 4; 'tau1', 1e-2; 'tau2', 1e-1; 'line_search', 1; 'solver', 0};
[X, obj, data] = msdp_al_engine('generic', struct('n', K.s, 'At', At, 'b', b, 'c', c), options, defaults);
end
