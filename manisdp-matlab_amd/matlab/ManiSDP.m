% Drop-in for the generic entry point src/primal/ManiSDP.m (Min <C,X> s.t. A(X) = b, X >= 0) on the GPU
% library: the Riemannian trust-region solve on euclideanfactory(n,p) runs in manisdp_mex('rtr', ...), the
% augmented-Lagrangian bookkeeping below stays in MATLAB.  Same signature, option names, defaults
% (ManiSDP.m:9-25), printed protocol and data fields as the reference.
function [X, obj, data] = ManiSDP(At, b, c, K, options)

def = {'p0',1; 'AL_maxiter',1000; 'gama',2; 'sigma0',1e-2; 'sigma_min',1e-1; 'sigma_max',1e7; 'tol',1e-8; ...
       'theta',1e-2; 'delta',8; 'alpha',0.1; 'tolgradnorm',1e-8; 'TR_maxinner',20; 'TR_maxiter',4; ...
       'tau1',1e-2; 'tau2',1e-1; 'line_search',1; 'solver',0};
for k = 1:size(def,1)
    if ~isfield(options, def{k,1}); options.(def{k,1}) = def{k,2}; end
end
n = K.s;  m = size(b,1);
fprintf('ManiSDP is starting...\n');
fprintf('SDP size: n = %i, m = %i\n', n, m);

h = manisdp_mex('create_generic', At, b, c, n);
guard = onCleanup(@() manisdp_mex('destroy', h));
A = At';
normb = 1 + norm(b);
y = zeros(m,1);  sigma = options.sigma0;  p = options.p0;
if isfield(options, 'Y0'), Y = options.Y0; else, Y = randn(n, p); end   % euclideanfactory M.rand
U = [];
tr = struct('maxinner', options.TR_maxinner, 'maxiter', options.TR_maxiter, ...
            'tolgradnorm', options.tolgradnorm, 'unittrace', 1);       % 'unittrace' = n x p layout at the boundary
data.status = 0;  eta = inf;  last = [];
clock0 = tic;
for iter = 1:options.AL_maxiter
    manisdp_mex('set_multipliers', h, y, sigma);
    if ~isempty(U), Y = armijo_like(Y, U); end
    [Y, info] = manisdp_mex('rtr', h, Y, tr);              % trustregions(problem, Y, opts)
    gradnorm = info.gradnorm;
    X = Y*Y';  x = X(:);
    res = A*x - b;
    pinf = norm(res)/normb;
    y = y - sigma*res;
    obj = c'*x;
    S = reshape(c - At*y, n, n);
    [vS, dS] = eig(full(S), 'vector');
    dinf = max(0, -dS(1))/(1 + dS(end));
    by = b'*y;
    gap = abs(obj - by)/(abs(by) + abs(obj) + 1);
    [V, D, ~] = svd(Y, 'econ');
    e = diag(D);  if size(D,2) == 1, e = D(1); end
    r = sum(e >= options.theta*e(1));
    fprintf('Iter %d, obj:%0.8f, gap:%0.1e, pinf:%0.1e, dinf:%0.1e, gradnorm:%0.1e, r:%d, p:%d, sigma:%0.3f, time:%0.2fs\n', ...
            iter, obj, gap, pinf, dinf, gradnorm, r, p, sigma, toc(clock0));
    eta = max([pinf, gap, dinf]);
    if eta < options.tol, fprintf('Optimality is reached!\n'); break; end
    if mod(iter, 20) == 0
        if iter > 50 && all([gap, pinf, dinf] > last)
            data.status = 2;  fprintf('Slow progress!\n');  break;
        end
        last = [gap, pinf, dinf];
    end
    if r <= p - 1, Y = V(:,1:r)*diag(e(1:r)); p = r; end   % rank cut
    nne = min(sum(dS < 0), options.delta);                 % escape directions
    if options.line_search == 1
        U = [zeros(n, p) vS(:,1:nne)];
        Y = [Y zeros(n, nne)];
    else
        Y = [Y options.alpha*vS(:,1:nne)];
    end
    p = p + nne;
    if pinf < options.tau1*gradnorm
        sigma = max(sigma/options.gama, options.sigma_min);
    elseif pinf > options.tau2*gradnorm
        sigma = min(sigma*options.gama, options.sigma_max);
    end
end
data.X = X; data.y = y; data.S = S; data.gap = gap; data.pinf = pinf; data.dinf = dinf;
data.gradnorm = gradnorm; data.time = toc(clock0);
if data.status == 0 && eta > options.tol
    data.status = 1;  fprintf('Iteration maximum is reached!\n');
end
fprintf('ManiSDP: optimum = %0.8f, time = %0.2fs\n', obj, toc(clock0));

    % line_search of ManiSDP.m:130-140 with co() evaluated on the device at Y + alpha*U
    function nY = armijo_like(Y0, D0)
        a = 1;  f0 = manisdp_mex('linesearch_cost', h, Y0, D0, 0);
        for trial = 1:16
            if trial == 16 || manisdp_mex('linesearch_cost', h, Y0, D0, a) - f0 <= -1e-3, break; end
            a = 0.8*a;
        end
        nY = Y0 + a*D0;
    end
end
